import sys, time, torch, numpy as np
sys.path.insert(0,'.')
from oracle import cur_oracle as O
A = torch.randn(64, 100000).bfloat16().float(); S = A + 0.3*torch.randn(64,100000)
for nt in (1, 4, 8, 16, 32, 64):
    torch.set_num_threads(nt)
    t=time.perf_counter(); O.eval_approx_score_mat_for_all_topk(A, S, [1,10,50,100], 100); dt=time.perf_counter()-t
    print(nt, "threads:", 64/dt, "q/s", flush=True)
