"""companion of kernel_to_body.py: rewrites hipLaunchKernelGGL(NAME<..>, grid, block, smem, stream, args...) of the converted kernels
into SEMSEG_LAUNCH_BODY((NAME_body<..>), grid, smem, stream, args...).  usage: launch_to_body.py FILE NAME [NAME ...]"""
import sys


def split_top(s):
    parts, depth, cur = [], 0, ''
    for c in s:
        if c in '([{':
            depth += 1
        elif c in ')]}':
            depth -= 1
        if c == ',' and depth == 0:
            parts.append(cur)
            cur = ''
        else:
            cur += c
    parts.append(cur)
    return parts


def rewrite(src, names):
    out, i = '', 0
    key = 'hipLaunchKernelGGL('
    while True:
        j = src.find(key, i)
        if j < 0:
            return out + src[i:]
        k = j + len(key)
        depth, e = 1, k
        while depth:
            depth += (src[e] == '(') - (src[e] == ')')
            e += 1
        inner = src[k:e - 1]
        # the kernel name may hold commas inside <...>: take it up to the first top-level comma that follows its closing paren / '>'
        if inner.lstrip().startswith('('):
            d, p = 0, 0
            for p, c in enumerate(inner):
                d += (c == '(') - (c == ')')
                if d == 0 and c == ')':
                    break
            kname, rest = inner[:p + 1].strip()[1:-1], inner[p + 1:]
        else:
            p = inner.index(',')
            kname, rest = inner[:p].strip(), inner[p:]
        base = kname.split('<')[0].strip()
        if base not in names or src[max(0, j - 8):j].endswith('#define '):
            out += src[i:e]
            i = e
            continue
        args = split_top(rest.lstrip()[1:])                 # drop the comma after the kernel name
        grid, block, smem, st, call = args[0], args[1], args[2], args[3], args[4:]
        body = base + '_body' + kname[len(base):]
        out += src[i:j] + 'SEMSEG_LAUNCH_BODY((%s),%s,%s,%s,%s)' % (body, ' ' + grid.strip(), smem, st, ','.join(call))
        i = e


if __name__ == '__main__':
    path = sys.argv[1]
    s = open(path).read()
    s = rewrite(s, set(sys.argv[2:]))
    open(path, 'w').write(s)
