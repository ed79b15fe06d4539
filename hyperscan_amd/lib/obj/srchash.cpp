extern "C" const char *hsgpu_source_hash(void) { return "5b5e99dc0b613bf582d1b4f711134ef0"; }
