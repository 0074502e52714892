#!/usr/bin/env python3
"""SHA-256 of a kernel's machine code inside librt_amd.so -- ties a committed counter profile to the exact kernel
build it was taken from (bench.py: roofline.traffic_source; tools/collect_profiles.py writes it beside the counters).

The shared object carries its gfx950 code objects as clang offload bundles (`__CLANG_OFFLOAD_BUNDLE__`, uncompressed);
each device image is an ELF64 whose .symtab names the kernels.  The hash covers the FUNC symbols whose demangled-ish
name contains `needle` (e.g. "k_trace" -> every k_trace<...> instance of both precisions), sorted by name:
name + code bytes.  Pure Python, no ROCm tool needed.

usage: python tools/kernel_hash.py [needle] [path/to/librt_amd.so]
"""
import hashlib
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _device_images(blob):
    at = blob.find(MAGIC)
    while at >= 0:
        (n,) = struct.unpack_from("<Q", blob, at + len(MAGIC))
        o = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, o)
            o += 24
            triple = blob[o:o + tl]
            o += tl
            if size and b"amdgcn" in triple:
                yield blob[at + off: at + off + size]
        at = blob.find(MAGIC, at + len(MAGIC))


def _functions(elf):
    """(name, code bytes) of every FUNC symbol of an ELF64 little-endian image."""
    if elf[:4] != b"\x7fELF" or elf[4] != 2:
        return
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    for s in secs:
        if s[1] != 2:  # SHT_SYMTAB
            continue
        stroff = secs[s[6]][4]
        for k in range(s[5] // 24):
            name_i, info, _other, shndx, value, size = struct.unpack_from("<IBBHQQ", elf, s[4] + k * 24)
            if (info & 0xF) != 2 or size == 0 or shndx == 0 or shndx >= shnum:  # STT_FUNC, defined
                continue
            sec = secs[shndx]
            end = elf.index(b"\0", stroff + name_i)
            name = elf[stroff + name_i:end].decode(errors="replace")
            file_off = sec[4] + (value - sec[3])
            yield name, elf[file_off:file_off + size]


def kernel_hash(needle="k_trace", path=None):
    """{'sha256': hex, 'symbols': n, 'code_bytes': total} over the matching kernels of librt_amd.so (None if absent)."""
    path = path or os.environ.get("RT_AMD_LIB") or os.path.join(ROOT, "rustraytracer_amd", "librt_amd.so")
    if not os.path.exists(path):
        return None
    blob = open(path, "rb").read()
    found = {}
    for img in _device_images(blob):
        for name, code in _functions(img):
            if needle in name:
                found[name] = code
    if not found:
        return None
    h = hashlib.sha256()
    for name in sorted(found):
        h.update(name.encode())
        h.update(found[name])
    return {"sha256": h.hexdigest(), "symbols": len(found), "code_bytes": sum(len(c) for c in found.values())}


if __name__ == "__main__":
    import json
    print(json.dumps(kernel_hash(sys.argv[1] if len(sys.argv) > 1 else "k_trace", sys.argv[2] if len(sys.argv) > 2 else None)))
