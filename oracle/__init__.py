"""CPU oracle for the A-RecSys hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in the product package (a-recsys_amd/) may import this.  Allowed
importers: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

PARITY STATUS: "parity unpinned" for the floating-point graph.  The reference
(skywaLKer518/A-Recsys) expresses the path as a TensorFlow-1.0 op graph;
TensorFlow is neither vendored in the reference nor installable here, and the
reference has no tests/golden vectors.  oracle.ref_graph restates the graph
op-for-op from the reference's Python call sites plus documented TF-1.0 op
semantics; it is cross-checked in tests/ by (a) hand-computed known-answer
cases and (b) an independent torch-autograd witness.  The integer / sampler
helpers (utils/prepare_train.py, attributes/attribute.py) ARE importable, and
oracle.ref_host is pinned against golden vectors generated from them
(tests/golden/make_golden.py).
"""
