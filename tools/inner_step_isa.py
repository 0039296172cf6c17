"""Instruction counts of the traversal kernels' inner-node step, from the compiler's assembly (no GPU needed).

    python tools/inner_step_isa.py [-DNAME=VALUE ...] [--leaf] [--dump]

The kernels are bound by the vector instructions they issue per step (profiles/r04_issue_bound_ab.txt), so this is the
number to watch when the step is changed: per production traversal kernel, the basic blocks between the node fetch
(three global_load_dwordx4 in one block) and the end of the step, with their VALU / SALU / memory instruction counts."""
import collections
import os
import re
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import isa_stats


def blocks_of(text, symbol):
    start = text.index(symbol + ":")
    end = text.index("s_endpgm", start)
    out, cur = [], None
    for line in text[start:end].split("\n"):
        m = re.match(r"^(\.LBB\S+):", line)
        if m:
            cur = [m.group(1), []]
            out.append(cur)
        elif cur is not None and line.startswith("\t") and not line.startswith("\t.") and not line.startswith("\t;"):
            cur[1].append(line.strip())
    return out


def main():
    defs = [a[2:] if a.startswith("-D") else a for a in sys.argv[1:] if a.startswith("-D") or a.startswith("-f") or a.startswith("-m")]
    path = tempfile.mktemp(suffix=".s")
    isa_stats.stats(defs, keep=path)
    text = open(path).read()
    os.remove(path)
    for sym in re.findall(r"^(_ZN3crt1[45]k_trace_(?:closest|shadow)ILb[01]ELb0ELb[01]E\S*):", text, re.M):
        blocks = blocks_of(text, sym)
        # the step: from the block with the plane conversions up to the leaf phase's slot fetch (four dwordx4 in one block)
        first = next(i for i, (_, ins) in enumerate(blocks) if sum("v_cvt_f32_ubyte" in x for x in ins) >= 12)
        last = next((i for i, (_, ins) in enumerate(blocks) if i > first and sum("global_load_dwordx4" in x for x in ins) >= 4), len(blocks))
        valu = pk = spill = total = 0
        for _, ins in blocks[first:last]:
            total += len(ins)
            valu += sum(x.startswith("v_") for x in ins)
            pk += sum(x.startswith("v_pk_") for x in ins)
            # the HBM part of the stack: five address instructions per site, skipped by a branch unless a lane is that deep
            spill += 5 * sum(("global_store_dword " in x or "global_load_dword " in x) for x in ins)
        print(f"{isa_stats.demangle(sym).split('(')[0]:48s} inner step: {total} instructions in {last - first} blocks, {valu} VALU "
              f"({pk} packed), of which ~{spill} on the HBM-stack paths")
        if "--leaf" in sys.argv:
            # the leaf step: from the slot fetch (>= 4 global_load_dwordx4 in one block) to the block that ends the first copy of
            # triangle B's test (the compiler emits the step twice: once for a leaf's first slot, once for the loop over the
            # further slots of a multi-slot leaf, which the default builders never make). Per block: instructions, VALU,
            # selects (v_cndmask: slot_pick's vertex selection for triangle B), division sequences, loads.
            lf = last
            seen_div = 0
            print("    leaf step blocks (instructions / VALU / v_cndmask / division-sequence instructions / global loads / LDS):")
            tot = collections.Counter()
            for name, ins in blocks[lf:lf + 40]:
                c = dict(n=len(ins), valu=sum(x.startswith("v_") for x in ins), cnd=sum("v_cndmask" in x for x in ins),
                         div=sum(("v_div" in x or "v_rcp" in x) for x in ins), gld=sum("global_load" in x for x in ins),
                         ds=sum(x.startswith("ds_") for x in ins))
                if c["n"] >= 20:
                    print(f"      {name:12s} {c['n']:4d} {c['valu']:4d} {c['cnd']:3d} {c['div']:3d} {c['gld']:2d} {c['ds']:2d}")
                tot.update(c)
                seen_div += c["div"] > 0 and c["n"] >= 60
                if seen_div == 2:
                    break
            print(f"      total        {tot['n']:4d} {tot['valu']:4d} {tot['cnd']:3d} {tot['div']:3d} {tot['gld']:2d} {tot['ds']:2d}")
        if "--dump" in sys.argv:
            for name, ins in blocks[first:last]:
                print(name)
                print("\n".join("    " + x for x in ins))


if __name__ == "__main__":
    main()
