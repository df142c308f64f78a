#!/usr/bin/env python3
"""Build-time audit of a wavefront-specialised kernel's assembly (called by yogo_amd/csrc/build.sh on conv_bf16_ws3): an asm statement that
ends with LDS reads still in flight (its text ends `s_waitcnt lgkmcnt(N)`, N > 0, or without a wait) declares their destinations as
defined ("=&v" outputs) although the data has not landed.  LLVM's s_waitcnt insertion does not look inside inline asm, so a
compiler-generated instruction between that statement and the next one that reads, copies or spills one of those registers would see
stale data.  This script fails the build when such an instruction exists: for every statement it collects the destination registers of
the LAST N ds_read instructions and scans the compiler's instructions up to the next ;;#ASMSTART for any mention of them.
    python3 tools/audit_inflight.py FILE.s"""
import re
import sys


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def mentioned(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line):
        out |= regs_of(tok)
    return out


def main(path):
    lines = open(path).read().split("\n")
    bad, i, n_checked = [], 0, 0
    while i < len(lines):
        if ";;#ASMSTART" not in lines[i]:
            i += 1
            continue
        j = i + 1
        body = []
        while j < len(lines) and ";;#ASMEND" not in lines[j]:
            body.append(lines[j].strip())
            j += 1
        # DS reads in issue order and the waits of this statement
        inflight = []   # destinations of reads not yet waited for
        for ins in body:
            m = re.match(r"ds_read\w*\s+(v\[\d+:\d+\]|v\d+)", ins)
            if m:
                inflight.append(regs_of(m.group(1)))
                continue
            m = re.match(r"s_waitcnt\s+.*lgkmcnt\((\d+)\)", ins)
            if m:
                keep = int(m.group(1))
                inflight = inflight[len(inflight) - keep:] if keep else []
            elif re.match(r"s_waitcnt\s+(0|lgkmcnt\(0\))", ins):
                inflight = []
        k = j + 1
        if inflight:
            n_checked += 1
            hot = set().union(*inflight)
            while k < len(lines) and ";;#ASMSTART" not in lines[k]:
                ln = lines[k].strip()
                if ln and not ln.startswith((";", ".")) and not ln.endswith(":"):
                    if ln.startswith("s_waitcnt") and "lgkmcnt(0)" in ln:
                        break   # the compiler itself waited for every LDS operation
                    if mentioned(ln) & hot:
                        bad.append((k + 1, ln))
                k += 1
        i = j + 1
    if bad:
        print(f"{path}: {len(bad)} compiler instructions touch a register whose LDS read is still in flight:", file=sys.stderr)
        for no, ln in bad[:10]:
            print(f"  line {no}: {ln}", file=sys.stderr)
        return 1
    print(f"{path.split('/')[-1].split('-hip-')[0]} in-flight audit ok ({n_checked} statements end with LDS reads in flight; no compiler instruction touches their destinations)")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
