"""MI355X-native coupling-flow log_prob path behind the distrax.Bijector /
distrax.Transformed surface the reference's scripts use.

    import normalizing_flows_amd as distrax      # src/real_nvp_toy_densities.py:5
    from normalizing_flows_amd import real_nvp   # free functions of src/real_nvp.py

All arithmetic runs in libnf_hip.so (hand-written HIP for gfx950) through the C
ABI of include/nf_hip.h; importing this package does not need a GPU, computing
does.
"""
from ._lib import NFError
from .bijectors import Bijector, Chain, Inverse, MaskedCoupling, RationalQuadraticSpline, ScalarAffine
from .conditioner import (MLPConditioner, layers_from_haiku_params, layers_to_haiku_params, load_params,
                          make_conditioner, save_params)
from .distributions import (Distribution, Independent, MultivariateNormalDiag, Normal, Transformed,
                            Uniform)
from .flow import CouplingFlow, DeviceArray, FlowConfig
from .mnist import prepare_data
from .sharding import RcclComm, ShardedNLL, shard_rows

__all__ = [
    "prepare_data",
    "NFError", "Bijector", "Chain", "Inverse", "MaskedCoupling", "RationalQuadraticSpline", "ScalarAffine",
    "MLPConditioner", "layers_from_haiku_params", "layers_to_haiku_params", "load_params", "save_params",
    "make_conditioner", "Distribution", "Independent",
    "MultivariateNormalDiag", "Normal", "Transformed", "Uniform", "CouplingFlow", "DeviceArray",
    "FlowConfig", "RcclComm", "ShardedNLL", "shard_rows",
]
