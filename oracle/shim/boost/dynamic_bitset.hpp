/* Included by BloomDBG/RollingHash.h:15 but never used there. Intentionally empty. */
