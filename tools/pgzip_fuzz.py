"""Differential fuzz of the many-core gzip reader (fastq_utils_amd/host/fq_pgzip.h) against zlib's gzread.  CPU only.
  g++ -O2 -std=c++17 -o /tmp/pgzip_check tests/cxx/pgzip_check.cpp -lz -pthread
  python3 tools/pgzip_fuzz.py SEED N [/tmp/pgzip_check]
Random content (FASTQ of short and long reads, noise, repeats, text, mixtures), random deflate settings per member
(level, strategy, window, memory level, flush points), 1 - 30 members, then a cut, a flipped bit or bytes behind the
last member in half of the cases; random thread counts and chunk sizes.  Prints DIFF lines and keeps the files.
Round 4: seeds 1-3 and 11-13, 150 cases each: no difference (the five of seeds 1-3 were files whose magic the flip had
destroyed - gzread copies such a file through; the reader now does too).
Round 5: a quarter of the cases carry members in front whose trailer ends within a few bytes of the end of a round's
loaded window (the case the round-4 reader ended the file in, silently); seeds 21-26, 150 cases each."""
import os, random, struct, subprocess, sys, zlib
sys.path.insert(0, '/root/repo')
from tests.test_pgzip import fastq_text
seed0 = int(sys.argv[1]); n = int(sys.argv[2])
CHECK = sys.argv[3] if len(sys.argv) > 3 else '/tmp/pgzip_check'
OUT = os.path.dirname(CHECK) or '.'
bad = 0
for it in range(n):
    r = random.Random(seed0 * 100000 + it)
    kind = r.choice(['fastq', 'fastq', 'fastq_long', 'noise', 'rep', 'mixed', 'text'])
    size = r.choice([0, 1, 100, 5000, 70000, 400000, 1500000, 4000000])
    if kind == 'fastq': data = fastq_text(size // 250 + 1, it)[:size]
    elif kind == 'fastq_long': data = fastq_text(size // 6000 + 1, it, read_len=(2000, 5000))[:size]
    elif kind == 'noise': data = r.randbytes(size)
    elif kind == 'rep': data = (r.randbytes(r.randint(1, 300)) * (size // 50 + 1))[:size]
    elif kind == 'text': data = b''.join(r.choice([b'the ', b'quick ', b'brown ', b'fox\n', b'jumps ', b'ACGT']) for _ in range(size // 5))
    else: data = fastq_text(size // 500 + 1, it)[:size // 2] + r.randbytes(size // 4) + fastq_text(size // 1000 + 1, it + 1)
    # members
    parts = []
    o = 0
    nm = r.choice([1, 1, 1, 2, 5, 30])
    cuts = sorted(r.randrange(0, len(data) + 1) for _ in range(nm - 1))
    prev = 0
    raw = b''
    for c in cuts + [len(data)]:
        p = data[prev:c]; prev = c
        level = r.choice([0, 1, 1, 3, 6, 6, 9])
        strat = r.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])
        wbits = r.choice([15, 15, 15, 12, 9])
        mem = r.choice([8, 8, 9, 1, 4])
        co = zlib.compressobj(level, zlib.DEFLATED, -wbits, mem, strat)
        fe = r.choice([0, 0, 0, 1000, 30000, 200000])
        if fe:
            body = b''
            for q in range(0, len(p), fe):
                body += co.compress(p[q:q + fe]) + co.flush(r.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_SYNC_FLUSH]))
            body += co.flush()
        else:
            body = co.compress(p) + co.flush()
        hdr = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03"
        if r.random() < 0.2:
            hdr = b"\x1f\x8b\x08\x08\x00\x00\x00\x00\x00\x03" + b"name%d\x00" % it
        raw += hdr + body + struct.pack("<II", zlib.crc32(p) & 0xFFFFFFFF, len(p) & 0xFFFFFFFF)
    threads = r.choice([1, 2, 3, 4, 8]); chunk = r.choice([4096, 9000, 30000, 100000, 1 << 20])
    if r.random() < 0.25 and chunk * threads <= 400000:
        # a member in FRONT whose trailer ends within a few bytes of where a round's loaded window ends (stored blocks
        # of noise: the compressed length is known), or k windows of such members in front of that one
        lead = b''
        for k in range(r.choice([1, 1, 2, 3])):
            total = chunk * threads + r.randint(-14, 40)
            pay = r.randbytes(max(0, total - 18 - 5 * ((total - 18) // 65535 + 1)))
            co = zlib.compressobj(0, zlib.DEFLATED, -15)
            lead += b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + co.compress(pay) + co.flush() + struct.pack("<II", zlib.crc32(pay) & 0xFFFFFFFF, len(pay))
        raw = lead + raw
    what = r.choice(['ok', 'ok', 'ok', 'cut', 'flip', 'tail'])
    if what == 'cut' and len(raw) > 20: raw = raw[:r.randrange(1, len(raw))]
    elif what == 'flip' and len(raw) > 20:
        b = bytearray(raw); at = r.randrange(0, len(b)); b[at] ^= 1 << r.randrange(8); raw = bytes(b)
    elif what == 'tail': raw += r.choice([b'\x00' * 5, b'\x1f', b'\x1f\x8b', b'xyz' * 50, b'\x1f\x8b\x08\x00'])
    path = os.path.join(OUT, 'fuzz_%d.gz' % seed0)
    open(path, 'wb').write(raw)
    p = subprocess.run([CHECK, path, str(threads), str(chunk)], capture_output=True, text=True, timeout=600)
    if p.returncode != 0:
        bad += 1
        keep = os.path.join(OUT, 'fuzz_bad_%d_%d.gz' % (seed0, it))
        os.rename(path, keep)
        print('DIFF', seed0, it, kind, size, nm, what, threads, chunk, p.returncode, p.stdout.strip()[:300], flush=True)
print('done', seed0, n, 'bad', bad)
