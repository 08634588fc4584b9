"""cProfile of bench_extras.search_full (state_factored_search K=40 over 64 instructions on the fixture graphs)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speaker_follower_amd import bench_extras
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
conn = os.path.join(ROOT, 'tests', 'golden', 'connectivity')
dev = torch.device('cuda', 0)
print(bench_extras.search_full(conn, dev))
pr = cProfile.Profile()
pr.enable()
out = bench_extras.search_full(conn, dev)
pr.disable()
print(out)
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
