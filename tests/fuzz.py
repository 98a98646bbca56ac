"""Seeded FASTQ generators and mutators for the parity tests (host side, numpy only)."""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_fastq(rng, n, min_len=1, max_len=200, name_style="casava", hdr2_names=False, crlf=False,
               mate=1, rna=False, first_index=0):
    """n well-formed records.  Returns bytes."""
    out = []
    eol = b"\r\n" if crlf else b"\n"
    alphabet = np.frombuffer(b"ACGU" if rna else b"ACGT", dtype=np.uint8)
    for i in range(n):
        L = int(rng.integers(min_len, max_len + 1))
        seq = alphabet[rng.integers(0, 4, L)].tobytes()
        if L > 3 and rng.random() < 0.1:
            k = int(rng.integers(0, L))
            seq = seq[:k] + b"N" + seq[k + 1:]
        qual = (rng.integers(2, 41, L) + 33).astype(np.uint8).tobytes()
        idx = first_index + i
        if name_style == "casava":
            name = b"SYN:1:FC:%d:%d:%d:%d %d:N:0:ACGT" % (idx % 8 + 1, idx % 97, idx % 1013, idx, mate)
        elif name_style == "slash":
            name = b"read.%d/%d" % (idx, mate)
        elif name_style == "int":
            name = b"%d" % idx
        else:  # no suffix
            name = b"read_%d_x" % idx
        h2 = b"+" + (name if hdr2_names else b"")
        out.append(b"@" + name + eol + seq + eol + h2 + eol + qual + eol)
    return b"".join(out)


MUTATIONS = ["flip_seq", "flip_any", "del_byte", "ins_cr", "ins_nul", "drop_line", "dup_line",
             "bad_plus", "bad_at", "short_qual", "mix_ut", "empty_seq", "truncate", "high_qual",
             "hdr2_name", "empty_hdr", "strip_last_nl"]


def mutate(rng, img: bytes, kind: str) -> bytes:
    b = bytearray(img)
    if not b:
        return bytes(b)
    lines = img.split(b"\n")
    had_nl = img.endswith(b"\n")
    if had_nl:
        lines = lines[:-1]
    nrec = len(lines) // 4
    r = int(rng.integers(0, max(1, nrec)))

    def join(ls):
        return b"\n".join(ls) + (b"\n" if had_nl else b"")

    if kind == "flip_seq" and nrec:
        s = bytearray(lines[4 * r + 1])
        if s:
            s[int(rng.integers(0, len(s)))] = int(rng.choice(list(b"XZ+@ -*acgtnN.0123Uu")))
            lines[4 * r + 1] = bytes(s)
        return join(lines)
    if kind == "flip_any":
        b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        return bytes(b)
    if kind == "del_byte":
        del b[int(rng.integers(0, len(b)))]
        return bytes(b)
    if kind == "ins_cr":
        b.insert(int(rng.integers(0, len(b))), 13)
        return bytes(b)
    if kind == "ins_nul":
        b.insert(int(rng.integers(0, len(b))), 0)
        return bytes(b)
    if kind == "drop_line" and lines:
        del lines[int(rng.integers(0, len(lines)))]
        return join(lines)
    if kind == "dup_line" and lines:
        k = int(rng.integers(0, len(lines)))
        lines.insert(k, lines[k])
        return join(lines)
    if kind == "bad_plus" and nrec:
        lines[4 * r + 2] = b"-" + lines[4 * r + 2][1:]
        return join(lines)
    if kind == "bad_at" and nrec:
        lines[4 * r] = b"X" + lines[4 * r][1:]
        return join(lines)
    if kind == "short_qual" and nrec:
        lines[4 * r + 3] = lines[4 * r + 3][:-1]
        return join(lines)
    if kind == "mix_ut" and nrec:
        s = lines[4 * r + 1]
        lines[4 * r + 1] = s + b"U" if rng.random() < 0.5 else b"u" + s
        lines[4 * r + 3] = lines[4 * r + 3] + b"I"
        return join(lines)
    if kind == "empty_seq" and nrec:
        lines[4 * r + 1] = b""
        lines[4 * r + 3] = b""
        return join(lines)
    if kind == "truncate":
        return bytes(b[: int(rng.integers(0, len(b)))])
    if kind == "high_qual" and nrec:
        q = bytearray(lines[4 * r + 3])
        if q:
            q[int(rng.integers(0, len(q)))] = int(rng.choice([0x7E, 0x7F, 0x80, 0xFF, 0x20, 0x01]))
            lines[4 * r + 3] = bytes(q)
        return join(lines)
    if kind == "hdr2_name" and nrec:
        h = lines[4 * r][1:]
        choice = int(rng.integers(0, 4))
        if choice == 0:
            lines[4 * r + 2] = b"+" + h
        elif choice == 1:
            lines[4 * r + 2] = b"+" + h + b"x"
        elif choice == 2:
            lines[4 * r + 2] = b"+" + h[:-1]
        else:
            lines[4 * r + 2] = b"+" + h.split(b" ")[0]
        return join(lines)
    if kind == "empty_hdr" and nrec:
        lines[4 * r] = b"@" if rng.random() < 0.5 else b""
        return join(lines)
    if kind == "strip_last_nl":
        return img[:-1] if had_nl else img
    return img
