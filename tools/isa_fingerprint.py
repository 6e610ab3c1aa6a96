#!/usr/bin/env python3
"""Did a refactoring change the code of any kernel?  Compiles csrc/*.hip to gfx950 assembly with the library's own flags (build.flags_for) and
hashes every kernel's INSTRUCTION STREAM — labels renumbered, symbol names (template arguments, anonymous-namespace tags) stripped — so that
moving a kernel to a header or adding a defaulted template parameter compares equal while any change of the generated code does not.

    python tools/isa_fingerprint.py --write /tmp/before.json                   # on the old tree
    python tools/isa_fingerprint.py --check /tmp/before.json [--any-file]      # on the new tree: kernels whose code changed / vanished / appeared
    ... --map OLD=NEW        substring replacement on the SAVED mangled names when a signature changed (e.g. a new template parameter)

The hashes depend on the compiler build (ROCm 7.2.0 here): a working aid, not a golden fixture.  Round 6 used it to show that moving
igemm_pl_halo_kernel into csrc/halo_kernel.h and adding the DB / BM template parameters left all 31 kernels of conv_planes.hip untouched."""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unflow_amd"))


def kernels_of(asm_path):
    out, cur = {}, None
    for line in open(asm_path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur and re.match(r"^\s+(v_|s_|ds_|buffer_|global_|flat_|scratch_)", line):
            t = re.sub(r"_GLOBAL__N_\w+", "", line.strip())
            t = re.sub(r"\.LBB\d+_", ".LBB_", t)
            t = re.sub(r"_Z\w+", "SYM", t)
            out[cur].append(t)
    return {k: (hashlib.md5("\n".join(v).encode()).hexdigest(), len(v)) for k, v in out.items() if v}


def fingerprint():
    import build as B
    flags_for = getattr(B, "flags_for", lambda src: B.FLAGS)
    fp = {}
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for src in B.sources():
            s = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
            cmd = [B.HIPCC] + [f for f in flags_for(src) if f != "-fPIC"] + ["--cuda-device-only", "-S", src, "-o", s]
            procs.append((src, s, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
        for src, s, p in procs:
            if p.wait() != 0:
                raise SystemExit("hipcc failed on %s" % src)
            for k, v in kernels_of(s).items():
                fp["%s::%s" % (os.path.basename(src), k)] = v
    return fp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write")
    ap.add_argument("--check")
    ap.add_argument("--map", action="append", default=[], help="OLD=NEW substring replacement applied to the saved kernel names")
    ap.add_argument("--any-file", action="store_true", help="match kernels by name alone (a kernel that moved to another .hip file)")
    args = ap.parse_args()
    fp = fingerprint()
    if args.write:
        json.dump(fp, open(args.write, "w"), indent=0, sort_keys=True)
        print("%d kernels -> %s" % (len(fp), args.write))
        return 0
    if not args.check:
        for k, (h, n) in sorted(fp.items()):
            print(h, "%6d" % n, k)
        return 0
    old = json.load(open(args.check))
    for m in args.map:
        a, b = m.split("=", 1)
        old = {k.replace(a, b): v for k, v in old.items()}
    key = (lambda k: k.split("::", 1)[1]) if args.any_file else (lambda k: k)
    new = {key(k): v for k, v in fp.items()}
    old = {key(k): v for k, v in old.items()}
    changed = [k for k in old if k in new and tuple(new[k]) != tuple(old[k])]
    gone = [k for k in old if k not in new]
    added = [k for k in new if k not in old]
    for tag, lst in (("CHANGED", changed), ("GONE", gone), ("NEW", added)):
        for k in sorted(lst):
            print(tag, k, "" if tag != "CHANGED" else "%s -> %s instructions" % (old[k][1], new[k][1]))
    print("%d kernels before, %d now: %d changed, %d gone, %d new" % (len(old), len(new), len(changed), len(gone), len(added)))
    return 1 if changed or gone else 0


if __name__ == "__main__":
    sys.exit(main())
