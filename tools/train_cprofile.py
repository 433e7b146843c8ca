"""Host-side profile (cProfile, cumulative) of the training step of `bench.py --mode train`: where the Python / launch time goes."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import harness, train_step

tr = train_step.Trainer(harness.SHAPES['R'], 200, torch.device('cuda:0'), 1)
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 70)
