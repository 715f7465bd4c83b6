#!/usr/bin/env python
"""Lists the kernels of libtfc_hip.so with their private (scratch) segment size and register counts, read
from the AMDGPU metadata notes of the gfx950 code objects inside the library's clang offload bundles.
A spill inside a hot loop is the kind of regression that does not show up in any parity test.
Usage: python tools/check_scratch.py [path/to/libtfc_hip.so] [name substring ...]"""
import os
import struct
import sys

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def kernels(elf):
    """AMDGPU metadata (NT_AMDGPU_METADATA = 32, msgpack) of one ELF64 code object."""
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = elf[shoff + i * shentsize:shoff + (i + 1) * shentsize]
        stype, = struct.unpack_from("<I", sh, 4)
        if stype != 7:                       # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", sh, 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if ntype == 32 and name.startswith(b"AMDGPU"):
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    yield k


def scan(path):
    blob = open(path, "rb").read()
    out = {}
    for co in code_objects(blob):
        for k in kernels(co):
            out[k[".name"]] = {"scratch": k.get(".private_segment_fixed_size", 0), "vgpr": k.get(".vgpr_count", 0),
                               "agpr": k.get(".agpr_count", 0), "sgpr": k.get(".sgpr_count", 0),
                               "lds": k.get(".group_segment_fixed_size", 0)}
    return out


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(root, "compression_amd", "libtfc_hip.so")
    subs = [a for a in sys.argv[1:] if not os.path.exists(a)]
    table = scan(path)
    for name in sorted(table, key=lambda n: -table[n]["scratch"]):
        if subs and not any(s in name for s in subs):
            continue
        r = table[name]
        if r["scratch"] or subs:
            print(f"{r['scratch']:6d} B scratch  {r['vgpr']:3d} v {r['agpr']:3d} a {r['sgpr']:3d} s  {name[:110]}")
    print(f"{len(table)} kernels, {sum(1 for r in table.values() if r['scratch'])} with scratch")
