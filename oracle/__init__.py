"""CPU oracles (test infrastructure only; see each module header)."""
