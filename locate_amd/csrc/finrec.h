// Records of the batched end-of-backward finalisers (finalise.hip, norm.hip): passed BY VALUE in the kernel arguments.
#pragma once
#define FIN_MAX 32
struct FinRec {
    const void* p[8];
    long long l[2];
    int i[8];
};
struct FinBatch {
    FinRec r[FIN_MAX];
};
