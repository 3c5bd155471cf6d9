"""Copy what tests/experiments/r05_final.sh left under gpurun_out/ into profiles/ under this round's names (the judged copies)."""
import json, os, shutil
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G, P = os.path.join(REPO, 'gpurun_out'), os.path.join(REPO, 'profiles')
pairs = [('r05_z/bench_final.json', 'r05_z_bench.json'), ('r05_z/kernel_stats.csv', 'r05_z_kernel_stats.csv'),
         ('r05_z/pmc_traffic.json', 'r05_pmc_traffic.json'), ('r05_z/pmc_traffic_static.json', 'r05_pmc_traffic_static.json'),
         ('r05_z/pmc_summary.txt', 'r05_z_pmc_summary.txt'), ('r05_z/bench_u16.json', 'r05_z_bench_u16.json'),
         ('r05_z/bench_static.json', 'r05_z_bench_static.json'),
         ('r05_z/bench_static_default_chain.json', 'r05_z_bench_static_default_chain.json'),
         ('r05_z/bench_static_malvar.json', 'r05_z_bench_static_malvar.json'),
         ('r05_z/bench_e2e_microscopy.json', 'r05_z_bench_e2e_microscopy.json'),
         ('r05_z/bench_e2e_drone.json', 'r05_z_bench_e2e_drone.json'), ('r05_z/sizes.txt', 'r05_z_sizes.txt'),
         ('r05_z/static.txt', 'r05_z_static.txt'), ('r05_parity_gpu.tsv', 'r05_parity_gpu.tsv'),
         ('step_graph_rccl_x1.json', 'r05_step_graph_rccl_x1.json'),
         ('bench_rccl_x1_graph_trial.json', 'r05_bench_rccl_x1_graph_trial.json')]
for a, b in pairs:
    src = os.path.join(G, a)
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copyfile(src, os.path.join(P, b))
        print('copied', a, '->', b)
    else:
        print('MISSING', a)
d = json.loads(open(os.path.join(P, 'r05_z_bench.json')).read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline']['frac'], d['step_roofline']['frac'], d['roofline'].get('traffic'))
print({k.replace('r2l_launch_', ''): v['avg_us'] for k, v in d['kernels'].items()})
print('small', json.dumps(d.get('small_shapes'))[:900])
print('static', json.dumps(d.get('static_c3'))[:1500])
