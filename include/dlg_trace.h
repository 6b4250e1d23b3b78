/* dlg_trace.h -- per-trial record of a dog-leg solve.
 *
 * The reference's only per-iteration observable is its vnlog record
 * (dogleg.c:50-64: one record per trial step, accepted or rejected, emitted at
 * dogleg.c:1406,1435,1458).  This struct carries the same fields at full
 * precision plus the trial point and step vectors, so that the product driver
 * and the CPU oracle can be compared trial-by-trial (SURVEY.md 8d "Parity
 * check").  Both fill it through the same layout; it is test/diagnostic
 * plumbing, not part of the dogleg.h API.
 */
#ifndef DLG_TRACE_H
#define DLG_TRACE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { DLG_STEP_CAUCHY = 0, DLG_STEP_GAUSSNEWTON = 1, DLG_STEP_INTERPOLATED = 2 };

typedef struct
{
  int    iteration;            /* accepted-step count when the trial was made     */
  int    accepted;             /* 1 accepted, 0 rejected, 2 terminal un-applied    */
  int    step_type;            /* DLG_STEP_*                                       */
  int    did_step_to_edge;
  double norm2x_before;
  double norm2x_after;         /* NaN for the terminal un-applied trial            */
  double norm2_cauchy;         /* |updateCauchy|^2 at the "before" point           */
  double norm2_gn;             /* |updateGN|^2, NaN if GN was not computed (yet)   */
  double k_cauchy_to_gn;       /* NaN unless interpolated                          */
  double norm2_step;           /* as the reference reports it (dogleg.c:1200: the
                                  UNSCALED Cauchy length for an edge-clipped step) */
  double expected_improvement; /* -1 for the terminal trial (dogleg.c:1295)        */
  double observed_improvement;
  double rho;
  double trustregion_before;
  double trustregion_after;
  double lambda;               /* ctx->lambda after the trial                      */
} dlg_trial_t;

typedef struct
{
  int          capacity;       /* room for this many trials                        */
  int          ntrials;        /* filled by the solver (may exceed capacity: then
                                  only the first `capacity` are stored)            */
  int          ncallbacks;     /* number of user-callback evaluations              */
  int          nstate;
  dlg_trial_t* trials;         /* [capacity]                                       */
  double*      p_trial;        /* [capacity][nstate] point given to the callback   */
  double*      step;           /* [capacity][nstate] step_to_here of the trial     */
} dlg_trace_t;

#ifdef __cplusplus
}
#endif
#endif
