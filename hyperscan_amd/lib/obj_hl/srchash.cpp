extern "C" const char *hsgpu_source_hash(void) { return "ab40e5324aeee7b79ea52283b698568d"; }
