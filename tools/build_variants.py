#!/usr/bin/env python3
"""Build timing-only variants of the library into ab_libs/: build_variants.py [--only src.hip[,src2.hip]] name=-DFLAG1,-DFLAG2 ...
(--only: recompile just those sources with the flags and link the product objects of the others)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from sageattention_amd import _build
os.makedirs("ab_libs", exist_ok=True)
ONLY = None
if len(sys.argv) > 2 and sys.argv[1] == "--only":
    ONLY = sys.argv[2].split(",")
    del sys.argv[1:3]
def one(spec):
    name, _, flags = spec.partition("=")
    return _build.build_variant(os.path.join("ab_libs", name + ".so"), [f for f in flags.split(",") if f], only=ONLY)
with ThreadPoolExecutor(3) as ex:
    for r in ex.map(one, sys.argv[1:]):
        print(r)
