/*
 * ko_geo.c -- CPU ORACLE (test infrastructure, not product code; see ko.h).
 * Restates orthodrome.f90, constants.f90 and euler.f90.
 *
 * constants.f90:21-25 initialises BOTH pi (real) and pi_ (real*8) from the
 * default-real literal 3.14159265358979, so pi_ == (double)3.14159274f; the
 * same happens to earth_oblateness (a default-real division).  These are kept.
 */
#include "ko.h"
#include <math.h>

static const float  PI_F = 3.14159265358979f;                 /* constants.f90:21 */
static const double PI_D = (double)3.14159265358979f;         /* constants.f90:22 */
static const float  EARTHRADIUS = 6371.f * 1000.f;            /* constants.f90:23 */
#define EARTHRADIUS_EQUATOR ((float)(6378.14f * 1000.f))      /* constants.f90:24 */
#define EARTH_OBLATENESS ((double)(1.f / 298.257223563f))     /* constants.f90:25 */

/* orthodrome.f90:331-338; 2./360.*pi is a default-real constant expression */
double ko_d2r_d(double deg) { return (double)((2.f / 360.f) * PI_F) * deg; }
/* orthodrome.f90:313-320 */
float ko_d2r_r(float deg) { return ((2.f / 360.f) * PI_F) * deg; }

static double clip(double x, double mi, double ma) { return fmin(fmax(mi, x), ma); }      /* :158-164 */
static double wrap(double x, double mi, double ma) { return x - floor((x - mi) / (ma - mi)) * (ma - mi); } /* :166-170 */

/* orthodrome.f90:284-293 */
static double cosdelta(ko_geo a, ko_geo b)
{
    return sin(a.lat) * sin(b.lat) + cos(a.lat) * cos(b.lat) * cos(b.lon - a.lon);
}

/* orthodrome.f90:245-265 */
void ko_azibazi(ko_geo a, ko_geo b, double *azimuth, double *backazimuth)
{
    double t = cos(a.lat) * cos(b.lat) * sin(b.lon - a.lon);
    double sb = sin(b.lat);
    double sa = sin(a.lat);
    double cd = cosdelta(a, b);
    *azimuth = atan2(t, sb - sa * cd);
    *backazimuth = atan2(-t, sa - sb * cd);
}

/* orthodrome.f90:193-229 */
double ko_distance_accurate50m(ko_geo a, ko_geo b)
{
    double f = (a.lat + b.lat) / 2.;
    double g = (a.lat - b.lat) / 2.;
    double l = (a.lon - b.lon) / 2.;
    double sg = sin(g), cl = cos(l), cf = cos(f), sl = sin(l), cg = cos(g), sf = sin(f);
    double s = (sg * sg) * (cl * cl) + (cf * cf) * (sl * sl);
    double c = (cg * cg) * (cl * cl) + (sf * sf) * (sl * sl);
    double w = atan(sqrt(s / c));
    double r = sqrt(s * c) / w;
    double d = 2. * w * (double)EARTHRADIUS_EQUATOR;
    double h1 = (3. * r - 1.) / (2. * c);
    double h2 = (3. * r + 1.) / (2. * s);
    return d * (1. + EARTH_OBLATENESS * h1 * (sf * sf) * (cg * cg)
                   - EARTH_OBLATENESS * h2 * (cf * cf) * (sg * sg));
}

/* orthodrome.f90:77-156.  Both approximations are switched off (:67,:72): the
 * flat branch never runs (dist < -1 is false), the const-azimuth branch only
 * when r == 0 (dist/r = +Inf > huge). */
void ko_approx_differential_azidist(float delta_x, float delta_y, double azimuth, double backazimuth,
                                    double dist, double *new_azimuth, double *new_backazimuth,
                                    double *new_dist)
{
    const double max_distance_flat_approx = -1.;
    if (dist < max_distance_flat_approx) {
        double ndx = dist * cos(azimuth) - (double)delta_x;
        double ndy = dist * sin(azimuth) - (double)delta_y;
        *new_azimuth = atan2(ndy, ndx);
        *new_backazimuth = backazimuth + (*new_azimuth - azimuth);
        *new_dist = sqrt(ndx * ndx + ndy * ndy);
        return;
    }
    double r = (double)sqrtf(delta_x * delta_x + delta_y * delta_y);   /* default-real expression */
    if (dist / r > 1.79769313486231570815e308) {
        *new_azimuth = azimuth;
        *new_backazimuth = backazimuth;
        *new_dist = dist - ((double)delta_x * cos(azimuth) + (double)delta_y * sin(azimuth));
        return;
    }
    double a = r / (double)EARTHRADIUS;
    double b = dist / (double)EARTHRADIUS;
    double lambda = (double)atan2f(delta_y, delta_x);                  /* default-real atan2 */
    double gamma = azimuth - lambda;
    double c = acos(clip(cos(a) * cos(b) + sin(a) * sin(b) * cos(gamma), -1., 1.));
    double alpha = asin(clip(sin(a) * sin(gamma) / sin(c), -1., 1.));
    double beta = asin(clip(sin(b) * sin(gamma) / sin(c), -1., 1.));
    if (cos(a) - cos(b) * cos(c) < 0) {
        if (alpha > 0) alpha = PI_D - alpha; else alpha = -PI_D - alpha;
    }
    if (cos(b) - cos(a) * cos(c) < 0) {
        if (beta > 0) beta = PI_D - beta; else beta = -PI_D - beta;
    }
    *new_dist = c * (double)EARTHRADIUS;
    *new_backazimuth = wrap(backazimuth + alpha, -PI_D, PI_D);
    *new_azimuth = wrap(lambda - PI_D - beta, -PI_D, PI_D);
}

/* euler.f90:28-67; mat[row][col] */
void ko_init_euler(float alpha, float beta, float gamma, float mat[3][3])
{
    float ca = cosf(alpha), cb = cosf(beta), cg = cosf(gamma);
    float sa = sinf(alpha), sb = sinf(beta), sg = sinf(gamma);
    mat[0][0] = cb * cg - ca * sb * sg;
    mat[1][0] = sb * cg + ca * cb * sg;
    mat[2][0] = sa * sg;
    mat[0][1] = -cb * sg - ca * sb * cg;
    mat[1][1] = -sb * sg + ca * cb * cg;
    mat[2][1] = sa * cg;
    mat[0][2] = sa * sb;
    mat[1][2] = -sa * cb;
    mat[2][2] = ca;
}
