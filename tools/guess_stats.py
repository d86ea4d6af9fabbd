#!/usr/bin/env python3
"""Development aid: what the first pass's start-state guesses get wrong (run on the GPU box).  The first
pass alone (JDGPU_TEST_REPAIR_ROUNDS=0 leaves its guesses in place; the pass that follows is the host's
repair loop) against the true states of a converged decode: per stream, the share of wrong guesses, how
many of them differ in the data unit (component phase) only, and how many segments, decoded from a wrong
state, still end in the true exit state (the decode found back).  usage: guess_stats.py [c2|c3|c5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
files = bench.make_files(pkg, wl, 8)
dec = pkg.Decoder(0)
os.environ["JDGPU_FIRST_SPAN"] = "1"
for k, f in enumerate(files):
    b = dec.batch([f])
    b.decode()                                   # converged: true states
    t_in, t_exit = b.states(0)
    b.close()
    os.environ["JDGPU_TEST_REPAIR_ROUNDS"] = "0"
    os.environ["JDGPU_TEST_NO_RETRY"] = "1"
    b = dec.batch([f])
    try:
        b.decode()
    except Exception:
        pass
    g_in, g_exit = b.states(0)
    del os.environ["JDGPU_TEST_REPAIR_ROUNDS"], os.environ["JDGPU_TEST_NO_RETRY"]
    b.close()
    n = len(t_in)
    wrong = g_in != t_in
    unit_only = wrong & ((g_in & 0x1FFF) == (t_in & 0x1FFF))
    found_back = wrong & (g_exit == t_exit)
    d = pkg.parse(f)
    print(f"stream {k}: {d.width}x{d.height} H{d.comp[0].H}V{d.comp[0].V}, {n} segments: wrong guesses {wrong.sum()} ({wrong.mean():.0%}), "
          f"of them unit-only {unit_only.sum()} ({unit_only.sum() / max(1, wrong.sum()):.0%}), "
          f"decoded from a wrong state yet ending right {found_back.sum()} ({found_back.sum() / max(1, wrong.sum()):.0%})")
