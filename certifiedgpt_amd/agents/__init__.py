from .base import BaseAgent, setup_agent  # noqa: F401
from .registry import registry  # noqa: F401

__all__ = ["BaseAgent", "setup_agent", "registry"]
