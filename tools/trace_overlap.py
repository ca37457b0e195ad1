"""Lab: reads a rocprofv3 kernel-trace CSV and prints, over the busiest window, the GPU's busy time (union of kernel intervals), the
sum of kernel durations (sum / busy = average number of kernels in flight) and the heaviest kernels by summed duration.
python tools/trace_overlap.py TRACE.csv [SKIP_FRACTION]"""
import csv, sys, collections

import re


def short(k):
    """kernel name without the anonymous namespace, argument list and (for grouped launches) everything but the body's name"""
    m = re.search(r'grouped_kernel.*?N_1\d+([a-z0-9_]+_(?:body|entry))((?:I(?:Li\d+E|Lb[01]E)+E)?)', k)
    if m:
        return 'grouped ' + m.group(1) + (m.group(2) or '')
    k = k.replace('(anonymous namespace)::', '').replace('void ', '')
    return k.split('(')[0][:80]


rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')) for r in rows)
t0, t1 = iv[0][0], max(e for _, e, _, _ in iv)
cut = t0 + (t1 - t0) * skip  # (warm-up rounds of the lab come first)
iv = [x for x in iv if x[0] >= cut]
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(e for _, e, _, _ in iv) - iv[0][0]
total = sum(e - s for s, e, _, _ in iv)
print(f'{len(iv)} launches over {span / 1e6:.1f} ms: busy {busy / 1e6:.1f} ms ({busy / span:.2f}), kernel time {total / 1e6:.1f} ms, '
      f'{total / busy:.2f} kernels in flight while busy; queues: {len(set(q for *_, q in iv))}')
by = collections.defaultdict(lambda: [0, 0])
for s, e, k, _ in iv:
    k = short(k)
    by[k][0] += e - s
    by[k][1] += 1
for k, (d, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:22]:
    print(f'{d / 1e6:9.2f} ms {n:6d} x {d / n / 1e3:8.1f} us  {k}')
