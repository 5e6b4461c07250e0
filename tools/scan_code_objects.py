"""Disassemble the gfx950 code objects embedded in a HIP shared library and count instruction patterns.

  python tools/scan_code_objects.py gops_amd/libgops_hip.so 'v_pk_(mul|add|fma)_f32' s_swappc_b64

Used by tests/test_host_cpu.py to hold the shipped library to "no packed-fp32 VALU instructions, no device-side calls"
(DESIGN_LOG.md, round 4: the gfx950 packed-fp32 hazard behind the round-3 non-determinism)."""
import os, re, struct, subprocess, sys, tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def code_objects(lib_path):
    """gfx950 ELF images of every offload bundle in the library's .hip_fatbin section"""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fatbin.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat], check=True)
        data = open(fat, "rb").read()
    magic, out, pos = b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        (nb,) = struct.unpack_from("<Q", data, i + 24)
        o = i + 32
        for _ in range(nb):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            triple = data[o + 24:o + 24 + tl].decode()
            o += 24 + tl
            if "gfx950" in triple and size:
                out.append(data[i + off:i + off + size])
        pos = i + 24
    return out


def scan(lib_path, patterns):
    """{pattern: count} over the disassembly of all device code, and the number of kernels seen"""
    counts, kernels = {p: 0 for p in patterns}, 0
    regs = {p: re.compile(p) for p in patterns}
    for img in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            proc = subprocess.Popen([OBJDUMP, "-d", f.name], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            for line in proc.stdout:
                if line.endswith(">:\n"):
                    kernels += 1
                    continue
                for p, r in regs.items():
                    if r.search(line):
                        counts[p] += 1
            proc.wait()
    return counts, kernels


if __name__ == "__main__":
    c, k = scan(sys.argv[1], sys.argv[2:] or [r"v_pk_(mul|add|fma)_f32", "s_swappc_b64"])
    print(f"{k} functions", c)
