import cProfile, pstats, sys, os, runpy, torch
sys.argv = ['pp_bench.py', '--steps', '3']
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pp_bench.py'))
fwd = ns['fwd']
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(3):
    fwd()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
