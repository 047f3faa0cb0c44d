"""Minimal registry with the reference's agent-plugin contract (common/registry.py:54-80 `register_agent`,
:207-212 `get_agent_class`, :112-133 generic `register`/`get`).  Only what the certify/predict path needs."""


class Registry:
    mapping = {"agent_name_mapping": {}, "state": {}}

    @classmethod
    def register_agent(cls, name):
        def wrap(agent_cls):
            from .base import BaseAgent
            assert issubclass(agent_cls, BaseAgent), "All agents must inherit BaseAgent class"
            if name in cls.mapping["agent_name_mapping"]:
                raise KeyError("Name '{}' already registered for {}.".format(name, cls.mapping["agent_name_mapping"][name]))
            cls.mapping["agent_name_mapping"][name] = agent_cls
            return agent_cls
        return wrap

    @classmethod
    def get_agent_class(cls, name):
        return cls.mapping["agent_name_mapping"].get(name, None)

    @classmethod
    def register(cls, name, obj):
        cls.mapping["state"][name] = obj

    @classmethod
    def get(cls, name, default=None):
        return cls.mapping["state"].get(name, default)

    @classmethod
    def get_configuration_class(cls, name):
        return cls.get(name)


registry = Registry()
