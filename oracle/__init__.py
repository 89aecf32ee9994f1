"""CPU oracle: test infrastructure only.  Never imported by upright_amd/."""
