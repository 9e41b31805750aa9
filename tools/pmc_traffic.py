"""Parses the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected in separate runs, with
--kernel-trace only) into profiles/pmc_hbm_traffic.json: per kernel symbol, average KB per launch and the
corrected HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B
requests at 64 B -- MI355X_MICROARCH.md 'HBM'; WRITE_SIZE uncalibrated).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out_f -o f --output-format csv -- python bench.py --steps 2 --warmup 1 --no-graph --profile-steps 0 --cpu-seconds 0
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out_w -o w --output-format csv -- python bench.py ... (same)
    python tools/pmc_traffic.py out_f/f_counter_collection.csv out_w/w_counter_collection.csv profiles/pmc_hbm_traffic.json
"""
import collections
import csv
import json
import sys


def load(path, cname):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != cname:
            continue
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        d[name].append(float(r['Counter_Value']))
    return d


if __name__ == '__main__':
    f = load(sys.argv[1], 'FETCH_SIZE')
    w = load(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for k in sorted(f, key=lambda k: -sum(f[k])):
        if k.startswith('at::') or 'rocclr' in k or 'Cijk' in k:
            continue
        n = len(f[k])
        fs = sum(f[k]) / n
        ws = sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
        out[k] = {'launches': n, 'fetch_kb_avg': fs, 'write_kb_avg': ws,
                  'hbm_bytes_per_launch_corrected': (2 * fs + ws) * 1024, 'hbm_bytes_per_launch_raw': (fs + ws) * 1024}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print('wrote', sys.argv[3], len(out), 'kernels')
