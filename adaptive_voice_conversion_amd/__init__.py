"""MI355X-native AdaIN-VC forward/backward engine (drop-in for the hot path of
jjery2243542/adaptive_voice_conversion: model.AE + Solver.ae_step)."""
__version__ = "0.1.0"
