"""CPU oracle package (test infrastructure only; never imported by the product path)."""
