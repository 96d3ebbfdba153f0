"""CPU oracle for the A-RecSys hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in the product package (a-recsys_amd/) may import this.  Allowed
importers: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

PARITY STATUS: "parity unpinned" for the floating-point graph.  The reference
(skywaLKer518/A-Recsys) expresses the path as a TensorFlow-1.0 op graph;
TensorFlow is neither vendored in the reference nor installable here, and the
reference has no tests/golden vectors.  oracle.ref_graph restates the graph
op-for-op from the reference's Python call sites plus documented TF-1.0 op
semantics; it is cross-checked in tests/ by (a) hand-computed known-answer
cases, (b) an independent torch-autograd witness and (c), round 6, THIRD-PARTY
implementations of the two semantics SURVEY A.7 / A.8 call unverifiable:
torch.nn.LSTMCell (gates permuted i,j,f,o -> i,f,g,o, forget_bias folded into
the bias) for oracle.ref_lstm's forward + BPTT, and torch.optim.Adagrad(
initial_accumulator_value=0.1, eps=0) on dense and duplicate-index sparse
gradients for the optimiser (tests/test_oracle_cpu.py section 5).
oracle.tf1_witness upgrades both checks to the REAL TF-1 ops whenever a
TensorFlow with the v1 API is importable (it is not here).  The integer / sampler
helpers (utils/prepare_train.py, attributes/attribute.py) ARE importable, and
oracle.ref_host is pinned against golden vectors generated from them
(tests/golden/make_golden.py).
"""
