"""CPU oracle for the PCReg hot path -- TEST INFRASTRUCTURE ONLY (see pcreg_oracle.py)."""
