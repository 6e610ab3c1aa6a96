"""unflow_amd — MI355X-native UnFlow training step (hand-written HIP kernels behind the reference's
e2eflow.ops operator API).  See DESIGN.md."""
__version__ = "0.1.0"
