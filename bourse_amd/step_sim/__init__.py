"""Discrete-event runner and agents (counterpart of the reference's ``bourse.step_sim``)."""
from . import agents, runner
from .runner import run, run_many
