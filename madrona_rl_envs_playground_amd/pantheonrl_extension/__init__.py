"""Drop-in counterparts of the reference's ``pantheonrl_extension`` modules that
sit on the step hot path (vectorenv.py, vectorobservation.py, the VectorAgent
base of vectoragent.py).  Trainers, SB3/cleanrl agents and the process-based
AsyncVectorEnv of the reference are consumers of this API, not part of it."""
from .vectorobservation import VectorObservation  # noqa: F401
from .vectoragent import RandomVectorAgent, VectorAgent  # noqa: F401
from .vectorenv import DummyEnv, MadronaEnv, PlayerException, SyncVectorEnv, VectorMultiAgentEnv  # noqa: F401
