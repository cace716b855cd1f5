#!/usr/bin/env python3
"""Build timing-only variants of the library into ab_libs/: build_variants.py name=-DFLAG1,-DFLAG2 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from sageattention_amd import _build
os.makedirs("ab_libs", exist_ok=True)
def one(spec):
    name, _, flags = spec.partition("=")
    return _build.build_variant(os.path.join("ab_libs", name + ".so"), [f for f in flags.split(",") if f])
with ThreadPoolExecutor(3) as ex:
    for r in ex.map(one, sys.argv[1:]):
        print(r)
