"""Build product (usflows_amd) models from a spec + reference-layout state dict."""
from usflows_amd.synth import build_usflow as build_flow, make_base  # noqa: F401
