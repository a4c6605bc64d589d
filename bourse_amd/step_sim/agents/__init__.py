from .base_agent import BaseAgent, BaseNumpyAgent, InstructionArrays
from .random_agent import ManyBookNumpyRandomAgents, NumpyRandomAgents, RandomAgent
