"""Refresh the two records of profiles/pmc_traffic.json that the bench line quotes (8192^2 K = 4 stream kernel; 1024^2
resident kernel) from the CSVs tools/collect_profiles.sh left in profiles/.  python tools/pmc_to_json.py"""
import json


def load(path):
    d = {}
    for line in open(path):
        row = line.strip().split(',')          # the kernel name holds commas: count from the right
        d[row[-5]] = {'n': int(row[-4]), 'mean': float(row[-3]), 'min': float(row[-2]), 'max': float(row[-1])}
    return d


j = json.load(open('profiles/pmc_traffic.json'))
p = load('profiles/r03_pmc_8192.csv')
simd_cycles = 1024 * p['GRBM_GUI_ACTIVE']['mean'] / 8
fetch = p['FETCH_SIZE']['mean'] * 1024 * 2
write = p['WRITE_SIZE']['mean'] * 1024
updates = 8192 * 8192 * 4
k = '8192x8192/math=exact/steps_per_launch=4/band=137/lane_cells=4'
valu_busy = 4 * p['SQ_ACTIVE_INST_VALU']['mean'] / simd_cycles
issue_busy = 4 * p['SQ_ACTIVE_INST_ANY']['mean'] / simd_cycles
lane = p['SQ_INSTS_VALU']['mean'] * 64 / updates
share = lambda q, c, which: round(q[c][which] / q['SQ_WAVE_CYCLES'][which], 3)
j[k] = {
    "traffic_bytes_per_launch": round(fetch + write, 1), "fetch_bytes": round(fetch, 1), "write_bytes": round(write, 1),
    "valu_busy": round(valu_busy, 4), "issue_busy": round(issue_busy, 4),
    "valu_wave_instructions_per_launch": p['SQ_INSTS_VALU']['mean'], "salu_instructions_per_launch": p['SQ_INSTS_SALU']['mean'],
    "lane_instructions_per_update": round(lane, 1), "waves": p['SQ_WAVES']['mean'],
    "wave_cycles_share": {"issuing (SQ_ACTIVE_INST_ANY)": share(p, 'SQ_ACTIVE_INST_ANY', 'mean'),
                          "parked at a wait (SQ_WAIT_ANY)": share(p, 'SQ_WAIT_ANY', 'mean'),
                          "issue stall (SQ_WAIT_INST_ANY)": share(p, 'SQ_WAIT_INST_ANY', 'mean')},
    "commit": "round 3, final (kernel stepk_pk<true,4,true,2,false,2>, band 137: one round of 2040 waves; blocked cells and the lid row by selects; warm-up iterations peeled off the row loop; 1.057 ms per 4-step launch in profiles/r03_8192_kernel_stats.csv)",
    "source": "profiles/r03_pmc_8192.csv (tools/pmc_grid.sh via tools/collect_profiles.sh)",
    "limiter_note": "instruction issue and DRAM traffic together: a VALU instruction issues in %.0f %% of all SIMD cycles and some instruction in %.0f %% (two waves per SIMD, 235 VGPRs), %.1f lane-instructions per update (round 2: 102.6), %.1f M scalar instructions per launch (round 2: 85 M); the memory side moves %.2f GB per launch = %.2f x compulsory (round 2: 1.20 x) at 5.2 TB/s, against the 5.4-5.8 TB/s this access pattern reaches at the memory controllers" % (100 * valu_busy, 100 * issue_busy, lane, p['SQ_INSTS_SALU']['mean'] / 1e6, (fetch + write) / 1e9, (fetch + write) / 4831838208.0)}
print(json.dumps(j[k], indent=1))

p = load('profiles/r03_pmc_1024_resident.csv')   # two launches: 200 warm-up steps, then the 2000 timed ones = the max
steps = 2000
simd_cycles = 1024 * p['GRBM_GUI_ACTIVE']['max'] / 8
valu_busy = 4 * p['SQ_ACTIVE_INST_VALU']['max'] / simd_cycles
issue_busy = 4 * p['SQ_ACTIVE_INST_ANY']['max'] / simd_cycles
valu_step = p['SQ_INSTS_VALU']['max'] / steps
salu_step = p['SQ_INSTS_SALU']['max'] / steps
lane = valu_step * 64 / (1024 * 1024)
looks = p['SQ_INSTS_VMEM_RD']['max'] / steps / 4096
j['1024x1024/resident'] = {
    "steps_of_the_profiled_launch": steps, "valu_busy": round(valu_busy, 4), "issue_busy": round(issue_busy, 4),
    "valu_wave_instructions_per_step": round(valu_step, 1), "salu_instructions_per_step": round(salu_step, 1),
    "lane_instructions_per_update": round(lane, 1), "waves": p['SQ_WAVES']['max'],
    "vmem_reads_per_wave_and_step": round(looks, 1),
    "write_bytes_per_step": round(p['WRITE_SIZE']['max'] * 1024 / steps, 1), "fetch_bytes_per_step": round(p['FETCH_SIZE']['max'] * 1024 * 2 / steps, 1),
    "wave_cycles_share": {"issuing (SQ_ACTIVE_INST_ANY)": share(p, 'SQ_ACTIVE_INST_ANY', 'max'),
                          "parked at a wait (SQ_WAIT_ANY)": share(p, 'SQ_WAIT_ANY', 'max'),
                          "issue stall (SQ_WAIT_INST_ANY)": share(p, 'SQ_WAIT_INST_ANY', 'max')},
    "commit": "round 3, final (kernel resident_band<1024,false,4>: 256 workgroups of 1024 threads, 4 waves per SIMD; lid row by selects; 3.83 us per step in profiles/r03_1024_resident_*)",
    "source": "profiles/r03_pmc_1024_resident.csv (the 2000-step launch of tools/pmc_grid.sh)",
    "limiter_note": "instruction issue: a VALU instruction issues in %.0f %% of all SIMD cycles -- %.1f lane-instructions per update, no redundant work -- and the rest of the issue slots goes to scalar, LDS and seam-granule instructions (%.0f scalar instructions and %.1f granule loads per wave and step); the lattice never leaves the registers: %.1f MB written and %.1f MB fetched beyond L2 per step (seams inside an XCD stay in its L2)" % (100 * valu_busy, lane, salu_step / 4096, looks, p['WRITE_SIZE']['max'] * 1024 / steps / 1e6, p['FETCH_SIZE']['max'] * 1024 * 2 / steps / 1e6)}
print(json.dumps(j['1024x1024/resident'], indent=1))
json.dump(j, open('profiles/pmc_traffic.json', 'w'), indent=1)
