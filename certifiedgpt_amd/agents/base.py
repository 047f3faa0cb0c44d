"""BaseAgent: the part of the reference's agents/base.py:22-155 contract that launch.py relies on
(`setup_agent` classmethod, `run`, `finalize`, `.config` from the registry)."""
from .registry import registry


class BaseAgent:
    def __init__(self):
        self.config = registry.get_configuration_class("configuration")   # agents/base.py:30
        self._model = None
        self._device = None

    @property
    def device(self):
        return self._device

    @property
    def model(self):
        return self._model

    @classmethod
    def setup_agent(cls, **kwargs):                                          # agents/base.py:153-155
        return cls()

    def run(self):
        raise NotImplementedError

    def finalize(self):
        raise NotImplementedError


def setup_agent(config):
    """agents/__init__.py:14-21."""
    assert "agent" in config["run"], "Agent name must be provided."
    agent_name = config["run"]["agent"]
    agent_cls = registry.get_agent_class(agent_name)
    assert agent_cls is not None, "Agent {} not properly registered.".format(agent_name)
    return agent_cls.setup_agent(cfg=config)
