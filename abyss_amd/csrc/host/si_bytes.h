// si_bytes.h -- SIToBytes (Common/StringUtil.h:181-219) as both host binaries need it for `-b`:
// a number, optionally followed by ONE unit character k / m / g in either case (powers of 1024);
// the product is rounded UP to whole bytes; anything else after the number is an error.
#pragma once
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdlib>

static inline bool si_to_bytes(const char* s, uint64_t* out)
{
	// (operator>>(double) takes a decimal floating-point literal; strtod would also take hex,
	// "inf" and "nan", which a stream rejects)
	const char* q = s;
	if (*q == '+' || *q == '-') q++;
	if (!(isdigit((unsigned char)*q) || (*q == '.' && isdigit((unsigned char)q[1])))) return false;
	if (q[0] == '0' && (q[1] == 'x' || q[1] == 'X')) return false;
	char* end;
	double x = strtod(s, &end);
	if (end == s) return false;
	if (*end == 0) { *out = (uint64_t)ceil(x); return true; }
	if (end[1] != 0) return false; // unrecognised multi-character suffix
	switch (tolower((unsigned char)*end)) {
	case 'k': x *= (double)(1ull << 10); break;
	case 'm': x *= (double)(1ull << 20); break;
	case 'g': x *= (double)(1ull << 30); break;
	default: return false;
	}
	*out = (uint64_t)ceil(x);
	return true;
}
