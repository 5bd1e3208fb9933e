/* placeholder translation unit; filled in when the mask-head / reid oracles land */
int heads_oracle_version(void) { return 0; }
