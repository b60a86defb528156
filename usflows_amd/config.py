"""Every switch of the device path in ONE place (round 5: the ~20 ``USFLOWS_AMD_*`` reads that were scattered over the package and
the ``getenv`` calls of the kernels' host code are gone).

Environment (8 variables; each is read when it is asked for, so a test may set it at any time):

====================================  ========================  ==========================================================================
variable                              values (default first)    meaning
====================================  ========================  ==========================================================================
``USFLOWS_AMD_LIB``                   path                      the HIP library to load (A/B builds); default: ``csrc/libusflows_hip.so``
``USFLOWS_AMD_GEMM``                  bf16x3 | f16x2 | f32      arithmetic of the matrix-core GEMMs
``USFLOWS_AMD_PLANES``                auto | 1 | 0              inference on the planes pipeline (auto: from the measured cross-over)
``USFLOWS_AMD_MERGE_AFFINE``          auto | 1 | 0              compose runs of consecutive affine maps (auto: behind the accuracy probe)
``USFLOWS_AMD_TRAIN``                 device | composite        ``Flow.log_prob`` under autograd: HIP training path or the torch formulation
``USFLOWS_AMD_TRAIN_PLANES``          1 | 0                     large-batch training on the planes pipeline (0: the fp32-row path)
``USFLOWS_AMD_TRAIN_GRAPH``           1 | 0                     ``Flow.fit`` captures its step as a hipGraph
``USFLOWS_AMD_TUNE``                  name=value,...            presets for the tuning knobs below AND for the library's table (usf_set_tuning)
====================================  ========================  ==========================================================================

Tuning knobs (attributes of ``config``; tests and tools assign them -- ``monkeypatch.setattr(config, "save_hidden", False)`` --
or preset them through ``USFLOWS_AMD_TUNE``): A/B switches kept so that a measurement can be repeated, all on by default.
The library's own knobs (cross-overs, schedules; ``usflows_amd/csrc/usf_api.hip``) are set with ``config.set_lib(name, value)``.
"""
from __future__ import annotations

import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def _tune_env() -> dict:
    out = {}
    for item in os.environ.get("USFLOWS_AMD_TUNE", "").split(","):
        if "=" in item:
            k, v = item.split("=", 1)
            out[k.strip()] = v.strip()
    return out


class Config:
    # ---- tuning knobs (Python side); every one documented where it is used --------------------------------------------------
    _KNOBS = dict(
        planes_min_rows=8192,      # engine: smallest batch of a planes plan
        base_in_epilogue=True,     # planes plans of Flow.log_prob: Laplace / Normal base density reduced by the last GEMM's epilogue
        tiny_coupling=True,        # launch-bound batches, tiny conditioners (the live flat configuration): a coupling layer as ONE launch each way
        engine_graph=False,        # engine: replay small inference batches as a hipGraph (measured no faster than usf_run_ops)
        save_hidden=True,          # training: conditioners' hidden activations kept by the forward instead of recomputed
        wgrad_planes=True,         # fp32-row training path: weight gradients from operand planes (usf_wgrad_planes_f32)
        wreduce_jobs=True,         # planes training: the weight gradients' reductions as one launch behind the layer loop
        fused_cbwd=True,           # fp32-row training path: the conditioner's backward as ONE fused launch
        fused_bias=True,           # bias gradients from the weight-gradient pass
        fit_prefetch=True,         # Flow.fit: next batch staged in pinned memory / uploaded under the running step
        image_train=True,          # image-shaped flows: HIP autograd functions (0: torch autograd + MIOpen)
        loop_list=True,            # image-shaped inference: the layer loop recorded as one op list
        loop_graph=True,           # ... or replayed as a hipGraph
        radial=True,               # RadialDistribution on usf_radial_logprob(_grad)_f32
        psum_jobs=True,            # small-batch conv weight gradients: last sums queued until the pass ends
        wgrad_jobs=True,           # ... and the weight-gradient launches themselves (one launch per tile shape when the pass ends)
        affine_prep=True,          # image flows: the affine blocks' parameter maps in one launch
        conv_res=True,             # conv kernel with MaskedCoupling's residual in its output stream
        pointwise=True,            # 1x1 convolutions on usf_pointwise_conv_f32
        batch_wplanes=True,        # image training, launch-bound batches: all convolution weights' planes from one launch per pass
        gated_tail_infer_max_pixels=1 << 14,   # inference: the same kernel instead of the one-thread-per-pixel pass up to this many pixels
        gated_tail_max_pixels=1 << 16,   # ... up to this many pixels (eight lanes share a pixel: made for few pixels)
        gated_tail=True,           # image training, few pixels: GatedConv's 1x1 conv + gate + ReLU + layer norm as one launch each way
    )

    def __init__(self):
        self.__dict__["_over"] = {}
        for k, v in self._KNOBS.items():
            self.__dict__[k] = v
        for k, v in _tune_env().items():
            if k in self._KNOBS:
                d = self._KNOBS[k]
                self.__dict__[k] = (v not in ("0", "false", "False", "")) if isinstance(d, bool) else type(d)(v)

    def __setattr__(self, name, value):
        if name not in self._KNOBS and name not in ("gemm_mode_default",):
            raise AttributeError(f"usflows_amd.config has no knob {name!r}")
        self.__dict__[name] = value

    # ---- the environment switches ----------------------------------------------------------------------------------------------
    @property
    def lib_path(self) -> str:
        return os.environ.get("USFLOWS_AMD_LIB", os.path.join(_HERE, "csrc", "libusflows_hip.so"))

    @property
    def gemm_mode(self) -> str:
        return os.environ.get("USFLOWS_AMD_GEMM", "bf16x3")

    @property
    def planes(self):
        """None = automatic, True / False = forced"""
        v = os.environ.get("USFLOWS_AMD_PLANES", "auto")
        return None if v == "auto" else v != "0"

    @property
    def merge_affine(self):
        """"auto" | True | False"""
        v = os.environ.get("USFLOWS_AMD_MERGE_AFFINE", "auto")
        return True if v == "1" else (False if v == "0" else "auto")

    @property
    def train_on_device(self) -> bool:
        return os.environ.get("USFLOWS_AMD_TRAIN", "device") != "composite"

    @property
    def train_planes(self) -> bool:
        return os.environ.get("USFLOWS_AMD_TRAIN_PLANES", "1") != "0"

    @property
    def train_graph(self) -> bool:
        return os.environ.get("USFLOWS_AMD_TRAIN_GRAPH", "1") != "0"

    # ---- the library's table -----------------------------------------------------------------------------------------------------
    @staticmethod
    def set_lib(name: str, value: int) -> None:
        from . import _ext
        _ext.check(_ext.load().usf_set_tuning(name.encode(), int(value)), "usf_set_tuning")

    @staticmethod
    def get_lib(name: str, default: int = 0) -> int:
        from . import _ext
        return int(_ext.load().usf_get_tuning(name.encode(), int(default)))


config = Config()
