#!/usr/bin/env python
"""tools/exp_dup_lockstep.sh output -> json: {base_ms, classes: {class: ms per pair added when every launch of it is issued twice}}.
bench.py reads the committed file for `roofline.timed_marginal` (the KPConv classes `fused` and `gather`)."""
import json
import re
import sys


def main(txt, source):
    base, dup = [], {}
    for line in open(txt):
        m = re.match(r'lockstep \S+ dup (\S+) -> ([\d.]+) pairs/s ([\d.]+) ms/pair', line)
        if m:
            (base if m.group(1) == 'none' else dup.setdefault(m.group(1), [])).append(float(m.group(3)))
    b = sum(base) / len(base)
    print(json.dumps({'source': source, 'base_ms_per_pair': b, 'base_pairs_per_s': 1e3 / b,
                      'classes': {k: max(sum(v) / len(v) - b, 0.0) for k, v in sorted(dup.items())}}, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
