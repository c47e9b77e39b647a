"""CPU oracle for the mask-render hot path -- TEST INFRASTRUCTURE ONLY (see oracle/ehr_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Parity status: unpinned (nvdiffrast is absent from /root/reference and the reference ships no golden vectors).
"""
