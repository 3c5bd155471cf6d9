"""Copy what tests/experiments/r04_final.sh left under gpurun_out/ into profiles/ under this round's names (the judged copies)."""
import json, os, shutil, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G, P = os.path.join(REPO, 'gpurun_out'), os.path.join(REPO, 'profiles')
pairs = [('r04/bench.json', 'r04_z_bench.json'), ('r04/kernel_stats.csv', 'r04_z_kernel_stats.csv'),
         ('r04/pmc_traffic.json', 'r04_pmc_traffic.json'), ('r04/pmc_summary.txt', 'r04_z_pmc_summary.txt'),
         ('r04/bench_u16.json', 'r04_z_bench_u16.json'), ('r04/bench_static.json', 'r04_z_bench_static.json'),
         ('r04/bench_static_default_chain.json', 'r04_z_bench_static_default_chain.json'),
         ('r04/bench_static_malvar.json', 'r04_z_bench_static_malvar.json'),
         ('r04/bench_e2e_microscopy.json', 'r04_z_bench_e2e_microscopy.json'),
         ('r04/bench_e2e_drone.json', 'r04_z_bench_e2e_drone.json'), ('r04/sizes.txt', 'r04_z_sizes.txt'),
         ('r04/static.txt', 'r04_z_static.txt'), ('r04/aux.txt', 'r04_z_aux.txt'), ('r04/epilogue.txt', 'r04_z_epilogue.txt'),
         ('r04/timeline_fwd.txt', 'r04_z_timeline_fwd.txt'), ('r04/timeline_fwd_noprio.txt', 'r04_z_timeline_fwd_noprio.txt'),
         ('r04/timeline_fwd_bwd.txt', 'r04_z_timeline_fwd_bwd.txt'),
         ('r04/timeline_fwd_bwd_noprio.txt', 'r04_z_timeline_fwd_bwd_noprio.txt'),
         ('r04_parity_gpu.tsv', 'r04_parity_gpu.tsv'), ('step_graph_rccl_x1.json', 'r04_step_graph_rccl_x1.json'),
         ('bench_gloo2_functional.json', 'r04_bench_gloo2_functional.json'),
         ('bench_rccl_x1_graph_trial.json', 'r04_bench_rccl_x1_graph_trial.json')]
for a, b in pairs:
    src = os.path.join(G, a)
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copyfile(src, os.path.join(P, b))
        print('copied', a, '->', b)
    else:
        print('MISSING', a)
d = json.loads(open(os.path.join(P, 'r04_z_bench.json')).read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline']['frac'], d['step_roofline']['frac'])
print({k.replace('r2l_launch_', ''): v['avg_us'] for k, v in d['kernels'].items()})
print('small', json.dumps(d.get('small_shapes'))[:900])
print('static', json.dumps(d.get('static_c3'))[:1500])
