/* Hand-written stand-in for the autoconf-generated config.h of the reference
 * (configure.ac:151-159 MAX_KMER/MAX_HASHES; AC_INIT name/version).
 * TEST INFRASTRUCTURE ONLY: used to compile the unmodified reference sources
 * from /root/reference into oracle/_ref/. Never linked into the product. */
#ifndef ORACLE_SHIM_CONFIG_H
#define ORACLE_SHIM_CONFIG_H 1
#define MAX_KMER 192
#define MAX_HASHES 32
#define PACKAGE_NAME "ABySS"
#define VERSION "2.3.10"
#define PACKAGE_BUGREPORT "abyss-users@bcgsc.ca"
#define HAVE_STD_HASH 1
#define HAVE_UNORDERED_SET 1
#define HAVE_UNORDERED_MAP 1
#define HAVE_FUNCTIONAL 1
#define HAVE_MEMORY 1
#define HAVE_STD_SHARED_PTR 1
#define HAVE_LIBDL 1
#define HAVE_DLFCN_H 1
#endif
