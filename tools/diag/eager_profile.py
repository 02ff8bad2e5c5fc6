"""Where does the HOST time of one eager stage-2 step go?   python tools/diag/eager_profile.py [batch]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream(dev))
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = bench.build_problem(b, dev, 1002)
step = bench.make_step(p)
for _ in range(20):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
