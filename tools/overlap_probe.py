"""Do two independent training steps on two HIP streams overlap on one MI355X?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from explainn_amd import ExplaiNN
from explainn_amd.engine import StepEngine
import bench
dev = torch.device("cuda", 0)
def make(U):
    torch.manual_seed(0)
    m = ExplaiNN(U, 19, 200, 1).to(dev).train(); m.validate_input = False
    return m, StepEngine(m, 1024)
x, y = bench.synthetic_batch(1024, 1, dev)
def run(engs, streams, steps=50):
    for _ in range(5):
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s): e.step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s): e.step(x, y)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e6
m300, e300 = make(300)
print("1 stream , 300 units: %.1f us/step" % run([e300], [torch.cuda.current_stream()]))
ma, ea = make(150); mb, eb = make(150)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
print("1 stream , 150 units: %.1f us/step" % run([ea], [s1]))
print("same stream, 2 x 150 units back to back: %.1f us/pair" % run([ea, eb], [s1, s1]))
print("2 streams, 2 x 150 units concurrently : %.1f us/pair" % run([ea, eb], [s1, s2]))
mc, ec = make(75); md, ed = make(75); s3, s4 = torch.cuda.Stream(), torch.cuda.Stream()
m4 = [make(75) for _ in range(4)]
print("4 streams, 4 x 75 units concurrently  : %.1f us/quad" % run([e for _, e in m4], [s1, s2, s3, s4]))
