"""Deterministic synthetic inputs for the BGZF/Mgzip hot path (bench slab + parity classes).

The reference's benchmark input is `bench-data/shakespeare.txt` catted 100 times (~550 MB;
README.md:166-167, benches/bench.rs:14-24); the file itself is absent from the checkout
(.MISSING_LARGE_BLOBS), so `text_slab()` builds a stand-in of the same shape: a 5.5 MiB
English-like base text repeated 100x = 576,716,800 bytes.

Everything derives from a vectorised splitmix64 stream so that the bytes are identical on
every numpy version (golden fixtures under tests/golden/ depend on that).
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def big_zeros(n):
    """n zero bytes backed by a pre-populated anonymous mapping.  Large numpy allocations are
    touched page by page on first use; in sandboxes with slow page faults (the build container:
    ~0.1 ms per 4 KiB page, i.e. 30 s per 256 MiB) populating the mapping up front is 300x faster."""
    import mmap
    if n < (8 << 20) or not hasattr(mmap, "MAP_POPULATE"):
        return np.zeros(n, dtype=np.uint8)
    m = mmap.mmap(-1, n, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | mmap.MAP_POPULATE)
    return np.frombuffer(m, dtype=np.uint8)


def splitmix64(seed, n, start=0):
    """n uint64 outputs of splitmix64 started at `seed` (vectorised: state_i = seed + (i+1)*G),
    beginning with output number `start`."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform01(seed, n):
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _sample_cdf(seed, n, probs):
    cdf = np.cumsum(np.asarray(probs, dtype=np.float64))
    cdf /= cdf[-1]
    return np.minimum(np.searchsorted(cdf, _uniform01(seed, n), side="right"), len(probs) - 1)


# English letter frequencies a..z (per mille, rounded)
_LETTER_FREQ = [82, 15, 28, 43, 127, 22, 20, 61, 70, 2, 8, 40, 24, 67, 75, 19, 1, 60, 63, 91, 28,
                10, 24, 2, 20, 1]
_SEPS = [b" ", b", ", b". ", b"\n"]
_SEP_P = [0.82, 0.08, 0.05, 0.05]


def english_like(n, seed=20250927):
    """n bytes of Zipf-distributed pseudo-English (4096-word vocabulary)."""
    if n == 0:
        return np.zeros(0, dtype=np.uint8)
    V = 4096
    # vocabulary: lengths ~ N(4.7, 2.2) clipped to [1, 12] via Box-Muller on the stream
    u1 = np.maximum(_uniform01(seed ^ 0x11, V), 1e-12)
    u2 = _uniform01(seed ^ 0x22, V)
    wl = np.clip(np.rint(4.7 + 2.2 * np.sqrt(-2.0 * np.log(u1)) * np.cos(2 * np.pi * u2)), 1,
                 12).astype(np.int64)
    letters = (_sample_cdf(seed ^ 0x33, V * 12, _LETTER_FREQ) + ord("a")).astype(np.uint8).reshape(
        V, 12)
    # rows = (word, separator) padded to 16 bytes
    table = np.zeros((V * 4, 16), dtype=np.uint8)
    rowlen = np.zeros(V * 4, dtype=np.int64)
    for s, sep in enumerate(_SEPS):
        rows = np.arange(V) * 4 + s
        table[rows, :12] = letters
        sb = np.frombuffer(sep, dtype=np.uint8)
        for k, b in enumerate(sb):
            table[rows, wl + k] = b
        rowlen[rows] = wl + len(sb)
    mean_len = 4.5 + 1.2
    out = np.empty(0, dtype=np.uint8)
    round_ = 0
    while out.size < n:
        m = int((n - out.size) / mean_len * 1.3) + 64
        ranks = _sample_cdf(seed ^ (0x1000 + round_), m, 1.0 / np.arange(1, V + 1))
        seps = _sample_cdf(seed ^ (0x2000 + round_), m, _SEP_P)
        rows = ranks * 4 + seps
        L = rowlen[rows]
        ends = np.cumsum(L)
        total = int(ends[-1])
        starts = ends - L
        within = np.arange(total, dtype=np.int64) - np.repeat(starts, L)
        chunk = table[np.repeat(rows, L), within]
        out = np.concatenate([out, chunk])
        round_ += 1
    return np.ascontiguousarray(out[:n])


def text_slab(total_bytes=576_716_800, base_bytes=5_767_168, seed=20250927):
    """The bench slab: a `base_bytes` pseudo-English text repeated to `total_bytes`
    (550 MiB = 5.5 MiB x 100, the shape of shakespeare.txt x 100)."""
    base = english_like(min(base_bytes, total_bytes), seed)
    out = big_zeros(total_bytes)
    for lo in range(0, total_bytes, base.size):
        hi = min(lo + base.size, total_bytes)
        out[lo:hi] = base[:hi - lo]
    return out


def dna(n, seed=1):
    return np.frombuffer(b"ACGT", dtype=np.uint8)[(splitmix64(seed, n) >> np.uint64(40)) &
                                                  np.uint64(3)].copy()


def fastq_like(n, seed=2):
    """FASTQ-shaped records: @id / bases / + / quals."""
    if n == 0:
        return np.zeros(0, dtype=np.uint8)
    recs = []
    size = 0
    r = splitmix64(seed, (n // 200 + 2) * 4)
    i = 0
    k = 0
    qual_alphabet = np.frombuffer(b"FFFFFF:,#", dtype=np.uint8)
    while size < n:
        ln = 100 + int(r[i] % np.uint64(51))
        hdr = ("@SRR%07d.%d %d/1\n" % (int(r[i + 1] % np.uint64(10_000_000)), k + 1, k + 1)).encode()
        bases = dna(ln, seed=int(r[i + 2] & np.uint64(0xFFFFFFFF)))
        q = qual_alphabet[(splitmix64(int(r[i + 3] & np.uint64(0xFFFFFFFF)), ln) >> np.uint64(33)) %
                          np.uint64(len(qual_alphabet))]
        rec = np.concatenate([np.frombuffer(hdr, dtype=np.uint8), bases,
                              np.frombuffer(b"\n+\n", dtype=np.uint8), q,
                              np.frombuffer(b"\n", dtype=np.uint8)])
        recs.append(rec)
        size += rec.size
        i += 4
        k += 1
    return np.ascontiguousarray(np.concatenate(recs)[:n])


def uniform_random(n, seed=3):
    return (splitmix64(seed, n) >> np.uint64(56)).astype(np.uint8)


def ascii_random(n, seed=8, start=0):
    """BASELINE config 3 bytes: 0x20 + (u8 % 95); bytes [start, start + n) of the stream."""
    if n <= (1 << 24):
        return (0x20 + ((splitmix64(seed, n, start) >> np.uint64(56)) % np.uint64(95))).astype(np.uint8)
    out = big_zeros(n)
    for lo in range(0, n, 1 << 24):  # 16 MiB pieces: the uint64 temporaries stay small
        m = min(1 << 24, n - lo)
        out[lo:lo + m] = ascii_random(m, seed, start + lo)
    return out


def byte_runs(n, seed=4):
    """Runs of a single byte, run lengths 1..600."""
    if n == 0:
        return np.zeros(0, dtype=np.uint8)
    m = n // 100 + 4
    r = splitmix64(seed, 2 * m)
    vals = (r[:m] >> np.uint64(56)).astype(np.uint8)
    lens = (r[m:] % np.uint64(600)).astype(np.int64) + 1
    out = np.repeat(vals, lens)
    while out.size < n:
        out = np.concatenate([out, out])
    return np.ascontiguousarray(out[:n])


def zeros(n, seed=0):
    return np.zeros(n, dtype=np.uint8)


def period2(n, seed=0):
    return np.resize(np.frombuffer(b"ab", dtype=np.uint8), n).copy()


def low_entropy_binary(n, seed=5):
    """Skewed 16-symbol alphabet (geometric-ish) -- stresses Huffman length limiting."""
    probs = 0.55 ** np.arange(16)
    sym = _sample_cdf(seed, n, probs)
    return (sym * 17).astype(np.uint8)


def mixed(n, seed=6):
    """random || text || random thirds."""
    a = n // 3
    b = n - 2 * a
    return np.concatenate([uniform_random(a, seed), english_like(b, seed + 1),
                           uniform_random(a, seed + 2)])


def repeated_phrases(n, seed=7):
    """Text with many long repeats (exercises nice_len / long matches / 258 cap)."""
    if n == 0:
        return np.zeros(0, dtype=np.uint8)
    base = english_like(max(n // 20, 64), seed)
    r = splitmix64(seed ^ 0x77, n // 40 + 8)
    parts = []
    size = 0
    i = 0
    while size < n:
        st = int(r[i] % np.uint64(base.size))
        ln = 20 + int((r[i] >> np.uint64(32)) % np.uint64(700))
        parts.append(base[st:st + ln])
        size += parts[-1].size
        i = (i + 1) % r.size
    return np.ascontiguousarray(np.concatenate(parts)[:n])


CLASSES = {
    "text": english_like,
    "dna": dna,
    "fastq": fastq_like,
    "runs": byte_runs,
    "zeros": zeros,
    "period2": period2,
    "random": uniform_random,
    "mixed": mixed,
    "lowent": low_entropy_binary,
    "repeats": repeated_phrases,
    "ascii": ascii_random,
}


def make(cls, n, seed=1):
    return CLASSES[cls](n, seed)
