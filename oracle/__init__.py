"""CPU oracle for the CompaCT hot path -- TEST INFRASTRUCTURE ONLY (see compact_oracle.c)."""
