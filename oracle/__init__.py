"""CPU oracle (test infrastructure only). See oracle/vibrato_oracle.c."""
