#!/usr/bin/env python3
"""hpgv_dev_commit sequences (MB, cumulative), one process each: which ones does this host's runtime accept?"""
import importlib, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    hpgv = importlib.import_module("hpg-variant_amd")
    e = hpgv.Engine(0)
    p = e.dev_reserve(3 << 30)
    out = []
    for mb in [int(x) for x in sys.argv[1].split(",")]:
        try:
            e.dev_commit(p, mb << 20); out.append("%d ok" % mb)
        except Exception as ex:
            out.append("%d FAILED (%s)" % (mb, str(ex)[-40:])); break
    print(sys.argv[1], "->", "; ".join(out))
    e.dev_release(p); e.close()
else:
    for seq in ("100,200,1024", "1024", "128,256,384,512,640,768,896,1024", "256,1024", "128,1024", "100,200,456", "100,200,300,1024", "2048", "100,2048"):
        subprocess.call([sys.executable, os.path.abspath(__file__), seq])
