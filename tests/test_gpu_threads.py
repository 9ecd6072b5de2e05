"""Two host threads, each with a context of its own, from the first launch on (a fresh process: every kernel's first launch — where launchers raise the dynamic-LDS limit
once per device, common.h PerDeviceOnce — is reached by both threads at about the same time).  One context = one calling thread is the contract (include/airfe.h); two
contexts on two threads must give each thread the results it gets alone."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys, threading
import numpy as np
sys.path.insert(0, os.getcwd())
from airslam_amd import api, synth, weights

S1 = os.path.join("tests", "golden", "plnet_s1.airfe")
sp, lg = weights.synthetic_plnet_s0(1234), weights.synthetic_lightglue(1234)
pairs = [synth.stereo_pair(480, 752, 3 * i) for i in range(4)]
start = threading.Barrier(2)
out = [None, None]

def work(t):
    start.wait()                                   # both threads create their context and reach every first launch together
    ctx = api.Context(superpoint=sp, lightglue=lg, plnet_s1=S1, max_batch=2, enc_chunk=2)
    res = []
    for rep in range(3):
        for (l, r) in pairs[2 * t:2 * t + 2]:
            res.append(ctx.stereo_keyframe(l, r))
    out[t] = res
    ctx.close()

th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
[x.start() for x in th]; [x.join() for x in th]
assert out[0] is not None and out[1] is not None, "a thread died"
# alone, afterwards, on one thread: the same calls
ctx = api.Context(superpoint=sp, lightglue=lg, plnet_s1=S1, max_batch=2, enc_chunk=2)
n = 0
for t in range(2):
    k = 0
    for rep in range(3):
        for (l, r) in pairs[2 * t:2 * t + 2]:
            ref = ctx.stereo_keyframe(l, r)
            got = out[t][k]; k += 1
            assert set(ref) == set(got)
            for key in ref:
                assert np.array_equal(np.asarray(ref[key]), np.asarray(got[key])), f"a thread's {key} differs from the single-threaded one"
                n += 1
assert n > 0
print("OK", n)
'''


def test_two_threads_two_contexts_from_the_first_launch():
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().startswith("OK"), (r.stdout[-1500:], r.stderr[-3000:])
