from .agent_copycat import AgentCopycat

agent_dict = {"agent_copycat": AgentCopycat}  # same key as uhc/agents/__init__.py:4-7 (AgentUHM is out of scope)
