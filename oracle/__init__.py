"""ORACLE -- test infrastructure only (see oracle/swmhd_oracle.c).  Never imported by swmhd_amd/."""
