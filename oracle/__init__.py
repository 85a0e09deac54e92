"""CPU oracle (test infrastructure only; see stabnet_oracle.py header). Parity unpinned."""
