#!/usr/bin/env python
"""Which co-running op makes layer1's 3x3 conv (conv_patch.hip) irreproducible?  (follow-up of exp_stream_race_ops.py)
The victim runs on two streams while ONE other op keeps four more streams busy; every victim output is compared bit
for bit with its single-stream reference."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0]] + sys.argv[1:]
os.environ.setdefault('RACE_REPS', '0')
exec(open(os.path.join(ROOT, 'scripts', 'exp_stream_race_ops.py')).read().split("refs = {}")[0])   # the job table
VICTIM = os.environ.get('RACE_VICTIM', 'l1.conv2')
N = int(os.environ.get('RACE_N', '600'))
ref = jobs[VICTIM]().clone()
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(6)]
for other in sorted(jobs):
    try:
        jobs[other]()
    except Exception:
        continue
    torch.cuda.synchronize()
    outs = []
    for i in range(N):
        for k, st in enumerate(streams):
            with torch.cuda.stream(st):
                if k < 2:
                    outs.append(jobs[VICTIM]())
                else:
                    jobs[other]()
        if len(outs) >= 200 or i == N - 1:
            torch.cuda.synchronize()
            bad = int(torch.stack([(o != ref).any() for o in outs]).sum())
            if bad:
                print('%s next to %s: %d mismatching outputs' % (VICTIM, other, bad), flush=True)
            outs = []
print('done: %s, %d launches next to each of %d ops' % (VICTIM, 2 * N, len(jobs)))
