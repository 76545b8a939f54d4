import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from dl_reference_models_amd import workloads as wl, _lib as L
from dl_reference_models_amd.vec_env import VecReferenceModel
b = 8192
cfg = wl.workload_config(wl.HEADLINE, list(range(b)))
env = VecReferenceModel(cfg); env.reset()
c = env.get_state()["counters"]; c[:, 0] = np.arange(b) % 100; env.set_state(counters=c)
acts = torch.randint(0, 5, (100, b, 8), dtype=torch.int8, device=env.device)
stream = torch.cuda.current_stream(); sptr = stream.cuda_stream
base, stride = acts.data_ptr(), b * 8
def plain(k, ptr=sptr):
    for t in range(k): env.step_raw(base + (t % 100) * stride, ptr, 1)
plain(100); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(); cs = torch.cuda.Stream()
with torch.cuda.graph(g, stream=cs): plain(20, torch.cuda.current_stream().cuda_stream)
for _ in range(5): g.replay()
torch.cuda.synchronize()
def med(f, n=30):
    xs = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); xs.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(xs)), float(np.min(xs))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); ev1.record()
print("sync only                 us", med(lambda: None))
print("2 event records           us", med(lambda: (ev0.record(stream), ev1.record(stream))))
print("graph(20) replay          us", med(lambda: g.replay()))
print("events + graph(20)        us", med(lambda: (ev0.record(stream), g.replay(), ev1.record(stream))))
def ev_ms():
    torch.cuda.synchronize(); ev0.record(stream); g.replay(); ev1.record(stream); torch.cuda.synchronize(); return ev0.elapsed_time(ev1) * 1e3
print("event time of graph(20)   us", float(np.median([ev_ms() for _ in range(30)])))
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=cs): plain(1, torch.cuda.current_stream().cuda_stream)
g1.replay(); torch.cuda.synchronize()
print("graph(1) replay           us", med(lambda: g1.replay()))
print("1 plain launch            us", med(lambda: plain(1)))
print("20 plain launches         us", med(lambda: plain(20)))
