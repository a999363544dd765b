"""One-off source transformation used in round 6 (kept for the record of how the kernels were converted): turns
    [template <...>] __global__ __launch_bounds__(N) void NAME(ARGS) { BODY }
into a body type of csrc/batch.h
    [template <...>] struct NAME_body { static constexpr int THREADS = N; static __device__ __forceinline__ void run(const U3 blockIdx,
                                        const U3 gridDim, ARGS) { BODY } };
leaving BODY untouched (the two leading parameters shadow the built-ins).  usage: kernel_to_body.py FILE NAME [NAME ...]"""
import re
import sys


def convert(src, name):
    m = re.search(r'__global__\s+__launch_bounds__\(([^)]*(?:\([^)]*\))?[^)]*)\)\s+void\s+' + re.escape(name) + r'\s*\(', src)
    assert m, name
    threads = m.group(1).strip()
    # matching ')' of the parameter list
    i = m.end()
    depth = 1
    while depth:
        c = src[i]
        depth += (c == '(') - (c == ')')
        i += 1
    args = src[m.end():i - 1]
    j = src.index('{', i)
    end = src.index('\n}\n', j)                  # the function's closing brace sits in column 0
    body = src[j + 1:end]
    head = 'struct %s_body {\n    static constexpr int THREADS = %s;\n    static __device__ __forceinline__ void run(const semseg_batch::U3 blockIdx, ' \
           'const semseg_batch::U3 gridDim,\n                                               %s) {' % (name, threads, args.strip())
    return src[:m.start()] + head + body + '\n    }\n};\n' + src[end + 3:]


if __name__ == '__main__':
    path = sys.argv[1]
    s = open(path).read()
    for n in sys.argv[2:]:
        s = convert(s, n)
    open(path, 'w').write(s)
