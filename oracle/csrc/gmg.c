/* placeholder translation unit: the geometric-multigrid oracle lands here. */
int orc_gmg_placeholder(void) { return 0; }
