"""CPU oracle of the PITA sampling path (test infrastructure only; see pita_oracle.py)."""
