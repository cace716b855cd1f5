import sys; sys.path.insert(0,'/root/repo')
from sageattention_amd import _build
V = {"nobar":["-DSAGE_EXP_NO_PROLOGUE_BARRIER"],
     "delay_nobar":["-DSAGE_EXP_DELAY_WAVE","-DSAGE_EXP_NO_PROLOGUE_BARRIER"],
     "delay_bar":["-DSAGE_EXP_DELAY_WAVE"],
     "ctemp":["-DSAGE_EXP_CTEMP"],
     "ctemp_nobar":["-DSAGE_EXP_CTEMP","-DSAGE_EXP_NO_PROLOGUE_BARRIER"]}
for n,f in V.items():
    print(_build.build_variant(f"/root/repo/ab_libs/lib_{n}.so", f), flush=True)
