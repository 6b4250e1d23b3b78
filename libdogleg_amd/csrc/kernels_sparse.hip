// placeholder until the sparse path lands
#include "dlg_internal.h"
int sparse_create(dlg_backend*) { return DLG_OK; }
size_t sparse_local_nnz(const dlg_backend* b) { return (size_t)b->nnz; }
void sparse_destroy(dlg_backend*) {}
int sparse_set_pattern(dlg_backend*, const int*, const int*) { dlg_set_error("sparse path not built yet"); return DLG_ERR_STATE; }
int sparse_eval(dlg_backend*, int) { dlg_set_error("sparse path not built yet"); return DLG_ERR_STATE; }
int sparse_norm2_Jv(dlg_backend*, int, const double*, double*) { dlg_set_error("sparse path not built yet"); return DLG_ERR_STATE; }
int sparse_factorize(dlg_backend*, int, double, int*) { dlg_set_error("sparse path not built yet"); return DLG_ERR_STATE; }
int sparse_solve(dlg_backend*, const double*, double*) { dlg_set_error("sparse path not built yet"); return DLG_ERR_STATE; }
extern "C" int dlg_sparse_stats(dlg_backend_t*, long*, long*, int*, int*, double*) { return DLG_ERR_STATE; }
