"""CPU oracle (test infrastructure only).  See oracle/oniris_oracle.py."""
