"""bourse_amd: many-book limit-order-book step simulator for AMD MI355X (gfx950).

Drop-in for the ``bourse_de::Env::step`` hot path of zombie-einstein/bourse: thousands of
independent books stepped in lockstep by hand-written HIP kernels (one wavefront per book)
behind a C ABI (include/bourse_amd.h).  See DESIGN.md / INTEGRATION.md.
"""
from . import _lib, core, data_processing, step_sim
from .compat import install_as_bourse, uninstall_bourse_alias
from ._lib import ACTION_MODIFY, BourseError, CapacityError, NoDeviceError
from .env import (MAX_PRICE, ManyBookEnv, ManyMarketEnv, MomentumAgent, MomentumParams, NoiseAgent, NoiseAgentParams,
                  RandomAgents, RandomMarketAgents, market_sim_runner, sim_runner)

__all__ = ["core", "step_sim", "data_processing", "install_as_bourse", "uninstall_bourse_alias", "ManyBookEnv", "RandomAgents", "NoiseAgent", "NoiseAgentParams", "MomentumAgent",
           "MomentumParams", "sim_runner", "ManyMarketEnv", "RandomMarketAgents", "market_sim_runner", "MAX_PRICE", "BourseError",
           "CapacityError", "NoDeviceError"]
