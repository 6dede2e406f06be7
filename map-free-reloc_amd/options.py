"""Declared kernel-selection options (A/B measurement and parity tests).  The C-ABI library reads no environment variable and neither
does this package: every switch that picks between two implementations of the same layer lives here, has a default, a closed set
of values, and is set either from the configuration (`cfg.HIP.<NAME>`, declared in config/default.py: unknown keys are rejected on
merge) through `apply_cfg(cfg)`, or directly by a tool / test (`options.set("CONV_KERNEL", "exact")`, `bench.py --hip-opt K=V`).

  SPLIT             'f16x2' (round 5: activations as two f16 terms, weights pre-scaled and packed as three; csrc/split_f16.h) | 'bf16x3'
                    (rounds 3-4: exact 3-way bf16 split of both operands): the arithmetic of the matrix-core kernels (linear layers,
                    3x3 convolutions); resolved when a weight is packed
  CONV              'wino' (own Winograd kernels) | 'miopen' (library convolution + own epilogue kernels): 3x3 layers of the matchers
  CONV_KERNEL       'auto' (nets/conv.py: f16x2 -> the direct halo-staged kernel, round 6; bf16x3 -> split / exact Winograd per layer shape) | 'direct'
                    (csrc/conv_direct.hip, f16x2 only) | 'split' (the operand-splitting Winograd kernel, arithmetic = SPLIT) | 'exact' (fp32 Winograd): which own 3x3 kernel
  FUSED_CONV1       SuperPoint conv1a + conv1b (+ ReLUs, max-pool) as ONE kernel that builds conv1b's input patches in LDS (True, f16x2 only) or as
                    two launches through the 6.4 GB intermediate (False); the same bits either way
  FUSED_CONV_RELU   SuperPoint conv1a through the fused first-layer kernel with ReLU folded (True / False)
  RPR_CONV          'hip' (own implicit-GEMM forward of the regression decoder's 3x3 convolutions) | 'miopen'
  RPR_CONV_BWD      'lib' (torch / MIOpen backward; the measured default) | 'hip' (own d input / d weight products)
  RPR_CONV_ORDER    'tap_inner' | 'tap_outer': K order of the forward / d input product
  RPR_WGRAD_SPLITS  K splits of the own d weight product (int >= 1)
  RPR_ENCODER_NHWC  the regression encoder's strided stages run on channels-last tensors (True / False): the library's NHWC kernels then
                    take their operands as they are instead of transposing every one (regression/encoder.py)
"""
_SPEC = {
    "SPLIT": ("f16x2", ("f16x2", "bf16x3")),
    "CONV": ("wino", ("wino", "miopen")),
    "CONV_KERNEL": ("auto", ("auto", "split", "exact", "direct")),
    "FUSED_CONV1": (True, (False, True)),
    "FUSED_CONV_RELU": (False, (False, True)),
    "RPR_CONV": ("hip", ("hip", "miopen")),
    "RPR_CONV_BWD": ("lib", ("lib", "hip")),
    "RPR_CONV_ORDER": ("tap_inner", ("tap_inner", "tap_outer")),
    "RPR_WGRAD_SPLITS": (16, None),
    "RPR_ENCODER_NHWC": (False, (False, True)),
}
_VALUES = {k: v[0] for k, v in _SPEC.items()}
_EXPLICIT = set()           # names set directly (tool / test / bench.py --hip-opt): a configuration that still carries the DEFAULT does not undo them


def names():
    return sorted(_SPEC)


def default(name):
    return _SPEC[name][0]


def get(name):
    return _VALUES[name]


def set(name, value):           # noqa: A001  (mirrors dict-like usage on purpose)
    if name not in _SPEC:
        raise KeyError(f"unknown HIP option {name!r}; known: {names()}")
    dflt, allowed = _SPEC[name]
    if isinstance(dflt, bool):
        if isinstance(value, str):
            value = value.lower() in ("1", "true", "yes", "on")
        value = bool(value)
    elif isinstance(dflt, int):
        value = int(value)
        if value < 1:
            raise ValueError(f"HIP option {name}: must be >= 1")
    if allowed is not None and value not in allowed:
        raise ValueError(f"HIP option {name}={value!r}: allowed {allowed}")
    _VALUES[name] = value
    _EXPLICIT.add(name)


class override:
    """with options.override(SPLIT="bf16x3"): ... -- the named options take the given values inside the block (objects that resolve an option when they
    are built, or on their first call, must be built AND first called inside it) and get their previous values and explicit-ness back afterwards"""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.saved = {k: (_VALUES[k], k in _EXPLICIT) for k in self.kv}
        for k, v in self.kv.items():
            set(k, v)
        return self

    def __exit__(self, *exc):
        for k, (v, ex) in self.saved.items():
            _VALUES[k] = v
            (_EXPLICIT.add if ex else _EXPLICIT.discard)(k)
        return False


def reset():
    for k, v in _SPEC.items():
        _VALUES[k] = v[0]
    _EXPLICIT.clear()


def apply_cfg(cfg):
    """copy every declared option that the configuration carries under HIP.*"""
    hip = cfg.HIP if "HIP" in cfg else {}
    for k in _SPEC:
        if k in hip:
            if k in _EXPLICIT and hip[k] == _SPEC[k][0]:
                continue                               # set directly, and the configuration only carries the declared default
            explicit = k in _EXPLICIT
            set(k, hip[k])
            if not explicit:
                _EXPLICIT.discard(k)
