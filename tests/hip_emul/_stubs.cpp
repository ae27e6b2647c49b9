extern "C" void pm_release_cached_memory(void) {}
