"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatements of the reference's hot path used only as the checker by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
multimodal_mvd_seg_amd/ imports this package.
"""
