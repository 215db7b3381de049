/* placeholder until the AVX2 restatement lands (keeps `make -C oracle` green) */
int zja_placeholder(void) { return 0; }
