"""TEST INFRASTRUCTURE ONLY: the parity oracle (C restatement of the reference CPU renderer) and the recipe that
compiles the unmodified reference.  Nothing under spath_amd/ imports this package."""
