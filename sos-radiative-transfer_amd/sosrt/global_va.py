"""Thresholds of the mu -> 0 treatments (SOS_Aer_global_va.py:5-7).  The file cache of
phase functions that the reference keeps in the same module is not part of the hot path."""
MU_THRESHOLD = 0.01
MU_EXTREME_THRESHOLD = 1e-8
MU_VERY_SMALL_THRESHOLD = 0.001
