"""Do three streams really run side by side?  Enqueue a long chain on the main stream, a chain on a side stream forked
from it by an event (as a Mixed block does), then ONE small kernel on a third stream that waits only for an event recorded
before all of it (as the text tower does).  Where in the main chain's time does the third stream's kernel run?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tumblr_emotions_amd import streams
streams.reserve()
main = torch.cuda.current_stream()
side, text = streams.get("side1"), streams.get("text")
x = torch.randn(4096, 4096, device="cuda")
y = torch.randn(4096, 4096, device="cuda")
small = torch.randn(1024, device="cuda")

def run(use_side, n=40):
    torch.cuda.synchronize()
    ready = torch.cuda.Event(); ready.record(main)
    t0 = torch.cuda.Event(enable_timing=True); t0.record(main)
    ev = torch.cuda.Event()
    for i in range(n):                       # ~n x 0.25 ms on main
        torch.mm(x, y)
        if use_side and i % 4 == 0:
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                torch.mm(x, y)
    t1 = torch.cuda.Event(enable_timing=True); t1.record(main)
    text.wait_event(ready)
    with torch.cuda.stream(text):
        a = torch.cuda.Event(enable_timing=True); a.record(text)
        small.add_(1.0)
        b = torch.cuda.Event(enable_timing=True); b.record(text)
    torch.cuda.synchronize()
    return t0.elapsed_time(t1), t0.elapsed_time(a), t0.elapsed_time(b)

for use_side in (False, True, False, True):
    tot, a, b = run(use_side)
    print("side chain %-5s: main chain %.2f ms; third stream's kernel ran at %.2f .. %.2f ms" % (use_side, tot, a, b))
