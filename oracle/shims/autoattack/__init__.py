"""Stand-in for fra31/auto-attack@a39220048b3c9f2cca9a4d3a54604793c68eca7e (own code).

Only `other_utils.{L0_norm,L1_norm,L2_norm,Logger}` is imported by the reference
(semseg/attacker.py:6); on the Linf path only Logger is reached.
"""
