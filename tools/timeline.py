"""Timeline of one steady-state step from a rocprofv3 --kernel-trace CSV: which queue each kernel ran on, when, what
ran beside it, and where the chip idled.
    python tools/timeline.py <kt_kernel_trace.csv> [--step k] [--list]
A step is found by its first kernel (fpn_gate_pack / fpn_gather, whichever comes first after the previous step's last
conv3x3 data gradient).  Prints per-queue busy time, the union busy time, idle gaps > 3 us and (with --list) every
kernel with its start offset, duration and queue."""
import csv, sys

path = sys.argv[1]
want = int(sys.argv[sys.argv.index('--step') + 1]) if '--step' in sys.argv else -2
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Queue_Id']), r['Kernel_Name']))
rows.sort()
# step boundaries: a gather kernel marks the neck forward of a new step
starts = [i for i, r in enumerate(rows) if 'fpn_gather_fwd_kernel' in r[3] or 'fpn_gather_gate_kernel' in r[3]]
bounds = []
for s in starts:
    # the step begins at the first kernel after the previous step's end: walk back over kernels that belong to this
    # step's prologue (gate pack / gate mfma on the side queue, zero fills)
    j = s
    while j > 0 and rows[j - 1][0] > rows[s][0] - 60_000 and not ('conv3x3' in rows[j - 1][3] or 'fpn_input_bwd' in rows[j - 1][3]):
        j -= 1
    bounds.append(j)
k = want if want >= 0 else len(bounds) + want
lo, hi = bounds[k], bounds[k + 1]
step = rows[lo:hi]
t0 = step[0][0]
t1 = max(r[1] for r in step)
print(f'step {k}: {len(step)} kernels, {(t1 - t0) / 1e3:.1f} us from first start to last end; next step starts at {(rows[hi][0] - t0) / 1e3:.1f} us')
queues = sorted({r[2] for r in step})
for q in queues:
    print(f'  queue {q}: busy {sum(r[1] - r[0] for r in step if r[2] == q) / 1e3:.1f} us in {sum(1 for r in step if r[2] == q)} kernels')
ev = sorted((r[0], r[1]) for r in step)
busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
for s, e in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((cur_e - t0, s - cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f'  union busy {busy / 1e3:.1f} us, idle {sum(g for _, g in gaps) / 1e3:.1f} us in {len(gaps)} gaps; both queues busy together '
      f'{(sum(r[1] - r[0] for r in step) - busy) / 1e3:.1f} us')
big = [(o, g) for o, g in gaps if g > 3000]
print('  gaps > 3 us:', ', '.join(f'{g / 1e3:.1f}us@{o / 1e3:.0f}' for o, g in big))
if '--list' in sys.argv:
    for s, e, q, n in step:
        n = n.replace('arfe::', '').replace('void ', '')
        print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {n[:100]}')
