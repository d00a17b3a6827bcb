"""CPU oracle for the JMAC hot path -- test infrastructure only (see jmac_oracle.py header)."""
