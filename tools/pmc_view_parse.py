"""Shares of SQ_WAVE_CYCLES per kernel from one tools/pmc_view.sh pass: parked (SQ_WAIT_ANY), issue-stalled (SQ_WAIT_INST_ANY),
VALU-active; LDS busy = SQ_LDS_IDX_ACTIVE / (launches x 256 CUs x duration x 2.4 GHz), conflicts as a share of it; MFMA pipe busy
= SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CU cycles).  python tools/pmc_view_parse.py gpurun_out/NAME_pm.txt"""
import re, collections, sys
rows=[l.split(None,5) for l in open(sys.argv[1]).read().splitlines()[1:]]
d=collections.defaultdict(dict)
for n,avg,tot,us,ctr,k in rows:
    k=re.sub(r"\(.*","",k).replace("void grafp::","")[:72]
    d[k][ctr]=float(tot); d[k]["us"]=float(us); d[k]["n"]=int(n)
g=lambda v,c: v.get(c,0.0)
for k,v in sorted(d.items(), key=lambda kv:-kv[1].get("SQ_WAVE_CYCLES",0)*1.0):
    wc=g(v,"SQ_WAVE_CYCLES") or 1.0
    cu_cycles = v["n"]*256*v["us"]*2400.0
    print("%-72s n=%4d %7.1fus wait=%.2f stall=%.2f valu=%.2f lds_busy=%.2f confl=%.2f mfma_busy=%.2f" % (
        k, v["n"], v["us"], g(v,"SQ_WAIT_ANY")/wc, g(v,"SQ_WAIT_INST_ANY")/wc, g(v,"SQ_ACTIVE_INST_VALU")/wc,
        g(v,"SQ_LDS_IDX_ACTIVE")/cu_cycles, g(v,"SQ_LDS_BANK_CONFLICT")/max(1.0,g(v,"SQ_LDS_IDX_ACTIVE")),
        g(v,"SQ_VALU_MFMA_BUSY_CYCLES")/(cu_cycles*4)))
