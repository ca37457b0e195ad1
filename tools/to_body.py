"""Developer aid (round 5): rewrites `__global__ void NAME(params) { body }` in a .hip file into
`__device__ NAME_body(const dim3 blockIdx, const dim3 gridDim, params) { body }` + a `__global__ NAME` that calls it (lockstep.h),
for the kernels named on the command line.   python tools/to_body.py FILE NAME [NAME ...]"""
import re, sys


def split_params(p):
    out, depth, cur = [], 0, ''
    for ch in p:
        if ch in '<([':
            depth += 1
        elif ch in '>)]':
            depth -= 1
        if ch == ',' and depth == 0:
            out.append(cur)
            cur = ''
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def convert(src, name):
    m = re.search(r'((?:template\s*<[^>]*>\s*\n)?)(static\s+)?__global__\s*((?:__launch_bounds__\([^)]*(?:\([^)]*\)[^)]*)*\)\s*)?)void\s+' + re.escape(name) + r'\s*\(', src)
    assert m, name
    tmpl, bounds = m.group(1), m.group(3)
    i = m.end()
    depth, j = 1, i
    while depth:
        depth += {'(': 1, ')': -1}.get(src[j], 0)
        j += 1
    params = src[i:j - 1]
    k = src.index('{', j)
    end = src.index('\n}\n', k) + 2  # kernels close at column 0
    body = src[k + 1:end - 1]
    names = [re.findall(r'[A-Za-z_][A-Za-z_0-9]*', p)[-1] for p in split_params(params)]
    targs = ''
    if tmpl:
        tp = split_params(re.search(r'<(.*)>', tmpl, re.S).group(1))
        targs = '<' + ', '.join(re.findall(r'[A-Za-z_][A-Za-z_0-9]*', t.split('=')[0])[-1] for t in tp) + '>'
    sep = ', ' if params.strip() else ''
    new = (f'{tmpl}__device__ __forceinline__ void {name}_body(const dim3 blockIdx, const dim3 gridDim{sep}{params}) {{\n'
           f'  (void)blockIdx; (void)gridDim;{body}}}\n'
           f'{tmpl}__global__ {bounds}void {name}({params}) {{ {name}_body{targs}(blockIdx, gridDim{sep}{", ".join(names)}); }}\n')
    return src[:m.start()] + new + src[end:]


if __name__ == '__main__':
    path = sys.argv[1]
    s = open(path).read()
    for n in sys.argv[2:]:
        s = convert(s, n)
    open(path, 'w').write(s)  # (only after every conversion succeeded)
