"""VGPR / SGPR / scratch / LDS of the kernels in libsdy_amd.so whose name contains the given substring (dev aid)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_code_objects as t

blob = open(t.LIB, "rb").read()
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for co in t._code_objects(blob):
    for k in t._kernels(co):
        if pat in k[".name"]:
            print(f"{k['.vgpr_count']:4d} vgpr {k['.sgpr_count']:4d} sgpr {k['.private_segment_fixed_size']:5d} B scratch "
                  f"{k['.group_segment_fixed_size']:7d} B lds  {k['.name']}")
