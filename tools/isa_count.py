"""Instruction mix of one kernel from hipcc's --save-temps assembly: python tools/isa_count.py <file.s> <kernel name substring> ..."""
import sys
from collections import Counter
s = open(sys.argv[1]).read()
for want in sys.argv[2:]:
    for line in s.splitlines():
        if want in line and not line.startswith((".", "\t", " ")) and ":" in line and line.split(":")[0].strip().isidentifier():
            k = line.split(":")[0]
            i = s.index("\n" + k + ":")
            j = s.find(".Lfunc_end", i)
            body = s[i:j if j > 0 else len(s)]
            ins = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
            c = Counter(ins)
            top = ("v_mad_u64_u32", "v_mul_lo_u32", "v_addc_co_u32_e32", "v_addc_co_u32_e64", "ds_read_b64", "ds_read2_b64", "ds_read_b128", "ds_write_b64", "ds_write2_b64", "ds_read_b32", "ds_write_b32",
                   "s_waitcnt", "s_barrier", "v_cndmask_b32_e32", "global_load_dwordx4", "global_store_dwordx4", "scratch_load_dword", "scratch_store_dword", "v_mov_b32_e32", "s_nop")
            print(k[:60], "total", len(ins), {x: c[x] for x in top if c[x]})
            break
