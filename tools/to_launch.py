"""Developer aid (round 5): rewrites hipLaunchKernelGGL(K, grid, dim3(T), lds, st, args...) of the kernels named on the command line into
::rdm::launch<K_body, K, T>(grid, lds, st, args...) (lockstep.h).   python tools/to_launch.py FILE NAME [NAME ...]"""
import re, sys


def split_top(s):
    out, depth, cur = [], 0, ''
    for ch in s:
        if ch in '([{':
            depth += 1
        elif ch in ')]}':
            depth -= 1
        if ch == ',' and depth == 0:
            out.append(cur)
            cur = ''
        else:
            cur += ch
    out.append(cur)
    return out


def convert(src, names):
    out, i = '', 0
    key = 'hipLaunchKernelGGL('
    while True:
        j = src.find(key, i)
        if j < 0:
            return out + src[i:]
        k = j + len(key)
        depth, e = 1, k
        while depth:
            depth += {'(': 1, ')': -1}.get(src[e], 0)
            e += 1
        inner = src[k:e - 1]
        # the kernel expression may contain commas inside <...>: split manually
        parts = split_top(inner)
        kern = parts[0].strip()
        rest = parts[1:]
        while kern.count('<') > kern.count('>'):  # template arguments were split at their commas
            kern += ',' + rest.pop(0)
        kern = kern.strip()
        bare = kern[1:-1].strip() if kern.startswith('(') and kern.endswith(')') else kern
        base = re.match(r'[A-Za-z_][A-Za-z_0-9]*', bare).group(0)
        if base not in names:
            out += src[i:e]
            i = e
            continue
        grid, block, lds, st = (x.strip() for x in rest[:4])
        args = ','.join(rest[4:])
        m = re.fullmatch(r'dim3\((.*)\)', block, re.S)
        assert m, (base, block)
        threads = m.group(1).strip()
        body = bare.replace(base, base + '_body', 1)
        out += src[i:j] + f'::rdm::launch<{body}, {bare}, {threads}>({grid}, {lds}, {st},{args})'
        i = e


if __name__ == '__main__':
    path = sys.argv[1]
    s = open(path).read()
    out = convert(s, set(sys.argv[2:]))
    open(path, 'w').write(out)
