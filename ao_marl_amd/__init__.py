"""ao_marl_amd: MI355X-native per-timestep AO environment hot path."""
__version__ = "0.1.0"
