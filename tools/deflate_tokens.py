"""The token list of a raw DEFLATE stream (pure Python, debugging aid): [(block, 'L', byte) | (block, 'M', length, distance)]
per DEFLATE block, with each block's type and position -- to see WHERE two streams that inflate to the same bytes part."""
LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXT = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193,
         12289, 16385, 24577]
DEXT = [0, 0, 0, 0] + [i // 2 for i in range(2, 28)]
ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


class Bits:
    def __init__(self, data):
        self.d, self.pos = data, 0

    def get(self, n):
        v = 0
        for i in range(n):
            v |= ((self.d[self.pos >> 3] >> (self.pos & 7)) & 1) << i
            self.pos += 1
        return v


def decoder(lens):
    codes, code = {}, 0
    for ln in range(1, 16):
        for sym, l in enumerate(lens):
            if l == ln:
                codes[(ln, code)] = sym
                code += 1
        code <<= 1

    def dec(b):
        c = 0
        for ln in range(1, 16):
            c = (c << 1) | b.get(1)
            if (ln, c) in codes:
                return codes[(ln, c)]
        raise ValueError("bad code")
    return dec


def tokens(raw):
    b, out, blocks, pos = Bits(raw), [], [], 0
    while True:
        final, typ = b.get(1), b.get(2)
        blocks.append((typ, pos, len(out)))
        if typ == 0:
            b.pos = (b.pos + 7) & ~7
            n = b.get(16)
            b.get(16)
            for _ in range(n):
                out.append((len(blocks) - 1, "L", b.get(8)))
            pos += n
        else:
            if typ == 1:
                ll = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8
                dl = [5] * 30
            else:
                hlit, hdist, hclen = b.get(5) + 257, b.get(5) + 1, b.get(4) + 4
                cl = [0] * 19
                for i in range(hclen):
                    cl[ORDER[i]] = b.get(3)
                cd, lens = decoder(cl), []
                while len(lens) < hlit + hdist:
                    s = cd(b)
                    if s < 16:
                        lens.append(s)
                    elif s == 16:
                        lens += [lens[-1]] * (3 + b.get(2))
                    elif s == 17:
                        lens += [0] * (3 + b.get(3))
                    else:
                        lens += [0] * (11 + b.get(7))
                ll, dl = lens[:hlit], lens[hlit:]
            ld, dd = decoder(ll), decoder(dl)
            while True:
                s = ld(b)
                if s < 256:
                    out.append((len(blocks) - 1, "L", s))
                    pos += 1
                elif s == 256:
                    break
                else:
                    ln = LBASE[s - 257] + b.get(LEXT[s - 257])
                    ds = dd(b)
                    out.append((len(blocks) - 1, "M", ln, DBASE[ds] + b.get(DEXT[ds])))
                    pos += ln
        if final:
            return out, blocks
