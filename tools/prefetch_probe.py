"""Timing experiment (not a product path): how much of the pack stage can hide on a side stream?
Baseline: train_step(x).  Probe: the batch of step i+1 is packed (explainn_stage_onehot) on a side
stream while step i runs, and train_step(NULL) uses the staged codes.  The input is the same every
step here, so the unsynchronised reuse of the pack buffers is harmless for a timing."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from explainn_amd import ExplaiNN, _lib
from explainn_amd.engine import StepEngine

U, k, L, T, B = 300, 19, 200, 1, 1024
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = ExplaiNN(U, k, L, T).to(dev).train()
eng = StepEngine(m, B, loss="binary")
idx = torch.randint(0, 4, (B, L), device=dev)
x = torch.nn.functional.one_hot(idx, 4).permute(0, 2, 1).float().contiguous()
y = (torch.rand(B, T, device=dev) > 0.5).float()
side = torch.cuda.Stream(dev)

def timed(fn, n=400, w=40):
    for i in range(w): fn(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

def base(i): eng.step(x, y, seed=i + 1)
print("train_step(x): %.4f ms" % timed(base))

ctx = m._context(B, dev); lib = ctx.lib; h = ctx.handle
main = torch.cuda.current_stream(dev)
ev_side, ev_main = torch.cuda.Event(), torch.cuda.Event()
def stage_on_side():
    ev_main.record(main); side.wait_event(ev_main)
    _lib.check(lib.explainn_stage_onehot(h, x.data_ptr(), B, C.c_void_p(side.cuda_stream)))
    ev_side.record(side)
stage_on_side()
def probe(i):
    main.wait_event(ev_side)
    _lib.check(lib.explainn_train_step(h, None, y.data_ptr(), B, C.byref(eng.ps), C.byref(eng.gs), eng.loss_kind,
               float(m.dropout_p), C.c_uint64(i + 1), 0, eng.logits.data_ptr(), eng.loss.data_ptr(),
               C.c_void_p(main.cuda_stream)))
    # the next batch's pack goes out right behind the step's first kernels: it overlaps the rest
    side.wait_event(ev_side)                      # (keeps the side stream in order; no main-stream dependency)
    _lib.check(lib.explainn_stage_onehot(h, x.data_ptr(), B, C.c_void_p(side.cuda_stream)))
    ev_side.record(side)
print("train_step(staged) + pack on a side stream: %.4f ms" % timed(probe))
def staged_only(i):
    _lib.check(lib.explainn_train_step(h, None, y.data_ptr(), B, C.byref(eng.ps), C.byref(eng.gs), eng.loss_kind,
               float(m.dropout_p), C.c_uint64(i + 1), 0, eng.logits.data_ptr(), eng.loss.data_ptr(),
               C.c_void_p(main.cuda_stream)))
print("train_step(staged), no pack at all (lower bound): %.4f ms" % timed(staged_only))
