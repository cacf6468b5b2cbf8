"""snarkvm_amd - MI355X (gfx950) backend for snarkVM's MSM / NTT hot path.

Drop-in scope: the accelerator boundary `snarkvm_algorithms_cuda::{NTT, polymul, msm}`
(reference algorithms/cuda/src/lib.rs:77-168) and the host types that call it
(`VariableBase`, `EvaluationDomain`, `PolyMultiplier`, `KZG10`).  See DESIGN.md.
"""
__version__ = "0.1.0"
