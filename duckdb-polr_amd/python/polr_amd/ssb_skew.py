"""SSB-skew data, generated -- on the host (numpy) or on the device (torch, no host copy) -- from ONE arithmetic.

The reference builds its skewed SSB instance by loading dbgen's .tbl files (not in the tree:
benchmark/ssb-skew/init/load.sql:74-78 reads them from PATHVAR) and then rewriting them with a fixed list of SQL
UPDATEs (load.sql:80-253).  Here the *base* tables are synthetic with the SSB schema's domains (uniform keys,
5 regions x 5 nations x 10 cities, 5 manufacturers x 5 categories x 40 brands, sizes 1..50, quantities 1..50,
calendar dates 1992-01-01 .. 1998-12-30, TPC-H-style sparse order keys, four lines per order = dbgen's mean) and the
UPDATEs are applied EXACTLY as written, in their order, to every row:

  customer   load.sql:81-82 (region/nation/city rewrite where c_custkey % 10 <> 0), :84-96 (+2 500 OCEANIA
             customers keyed above the base range), views c1/c2/c3 :98-100, lineorder.lo_custkey :102-132
  supplier   :139-140, views s1/s2/s3 :142-144, lineorder.lo_suppkey :146-178 (the third rule reads the customer
             the row points to AFTER the customer rules)
  part       :185-187
  date       lineorder.lo_orderdate := view_of_year[lo_orderkey % 365] by order-key band :190-245

The constants of load.sql are SF100-specific: `% 2942519`, `% 59981`, `% 2402`, `% 183979`, `% 787`, `% 191` are the
cardinalities of the views on the dbgen data, `400000000` / `80000000 * y` / `401000000` are fractions of the largest
order key (600 M at SF100).  On other data the same rules read: modulus = cardinality of that view here, order-key
thresholds = the same fractions of the largest order key (scale/100 x the constants).  Both are reported by
`params()` so that a run states what it used.

Every value is a pure function of (seed, row index): a partition [lo, hi) of lineorder generated on rank r of N, on the
device, equals rows [lo, hi) of the table generated anywhere else (what lets config 5 -- SF1000, 6 B rows -- exist
without ever being materialised on a host).  The arithmetic uses int64 two's-complement wrap-around and logical
shifts emulated with masks, which numpy and torch agree on bit for bit (tests/test_ssb_skew.py).
"""
import numpy as np

REGIONS = ["AFRICA", "AMERICA", "ASIA", "EUROPE", "MIDDLE EAST"]
R_AMERICA, R_ASIA, R_EUROPE, R_OCEANIA = 1, 2, 3, 5
# nation code = region * 5 + (0..4); the three nations load.sql names
N_UNITED_STATES = R_AMERICA * 5 + 4
N_CHINA = R_ASIA * 5 + 1
N_UNITED_KINGDOM = R_EUROPE * 5 + 3
N_OCEANIA0 = 25  # the 25 added nations ANDORRA .. ZIMBABWE: codes 25..49

_G = -7046029254386353131   # 0x9E3779B97F4A7C15 as int64
_M1 = -4658895280553007687  # 0xBF58476D1CE4E5B9
_M2 = -7723592293110705685  # 0x94D049BB133111EB
_SALT = 7145358350939287577  # 0x632BE59BD9B4E019


class _NP:
    """numpy flavour of the few array operations the generator needs (int64 everywhere)"""
    name = "numpy"

    @staticmethod
    def arange(lo, hi):
        return np.arange(lo, hi, dtype=np.int64)

    @staticmethod
    def where(c, a, b):
        return np.where(c, a, b)

    @staticmethod
    def const(v):
        return np.int64(v)

    @staticmethod
    def asarray(a):
        return np.asarray(a, dtype=np.int64)

    @staticmethod
    def take(table, idx):
        return table[idx]

    @staticmethod
    def cast(a, dtype):
        return a.astype(dtype)


class _Torch:
    name = "torch"

    def __init__(self, device):
        import torch
        self.torch = torch
        self.device = device

    def arange(self, lo, hi):
        return self.torch.arange(lo, hi, dtype=self.torch.int64, device=self.device)

    def where(self, c, a, b):
        t = self.torch
        if not t.is_tensor(a):
            a = t.full((), int(a), dtype=t.int64, device=self.device)
        if not t.is_tensor(b):
            b = t.full((), int(b), dtype=t.int64, device=self.device)
        return t.where(c, a, b)

    def const(self, v):
        return int(v)

    def asarray(self, a):
        return self.torch.as_tensor(np.asarray(a, dtype=np.int64), device=self.device)

    def take(self, table, idx):
        return table[idx]

    def cast(self, a, dtype):
        t = self.torch
        # torch has no uint32 arithmetic; the bit pattern of an int32 tensor IS the uint32 column the device reads
        m = {np.uint32: t.int32, np.int32: t.int32, np.uint16: t.int16, np.int16: t.int16}
        return a.to(m[dtype])


def _lsr(x, s):
    """logical shift right of an int64 array"""
    return (x >> s) & ((1 << (64 - s)) - 1)


def _mix(xp, x, salt):
    """splitmix64 finaliser over (x, salt) in wrap-around int64 arithmetic; returns 31 uniform bits (>= 0)"""
    z = (x + _wrap(salt * _SALT)) * _G
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    z = z ^ _lsr(z, 31)
    return _lsr(z, 33)


def _wrap(v):
    """Python int -> int64 two's complement"""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def date_table():
    """the 2 556 calendar days 1992-01-01 .. 1998-12-30 (SSB's date dimension): d_datekey = yyyymmdd, d_year"""
    keys, years = [], []
    mdays = [31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31]
    for y in range(1992, 1999):
        leap = y % 4 == 0
        for m in range(12):
            for d in range(mdays[m] + (1 if (leap and m == 1) else 0)):
                keys.append(y * 10000 + (m + 1) * 100 + d + 1)
                years.append(y)
    return np.asarray(keys[:2556], dtype=np.uint32), np.asarray(years[:2556], dtype=np.uint16)


def sizes(sf):
    """SSB cardinalities at scale factor sf (specification): lineorder 6 M x sf, customer 30 k x sf, supplier
    2 k x sf, part 200 k x floor(1 + log2 sf)"""
    sf = float(sf)
    n_p = int(200_000 * max(1, int(np.floor(1 + np.log2(max(sf, 1.0))))))
    return {"n_lo": int(6_000_000 * sf), "n_c": max(int(30_000 * sf), 1000), "n_s": max(int(2_000 * sf), 400),
            "n_p": n_p if sf >= 1 else max(int(200_000 * sf), 2000)}


class Instance:
    """dimension tables + views of one SSB-skew instance (host side, numpy; a few MB even at SF1000)"""

    def __init__(self, n_lo, n_c, n_s, n_p, seed=1337):
        self.n_lo, self.n_c, self.n_s, self.n_p, self.seed = int(n_lo), int(n_c), int(n_s), int(n_p), int(seed)
        xp = _NP
        s = _wrap(self.seed * 1000003)
        # ---- customer (base rows 1..n_c, then the 2 500 added OCEANIA rows) ------------------------------
        ck = np.arange(1, self.n_c + 1, dtype=np.int64)
        c_region = _mix(xp, ck, s + 1) % 5
        c_nation = c_region * 5 + _mix(xp, ck, s + 2) % 5
        c_digit = _mix(xp, ck, s + 3) % 10
        upd = (c_region != R_AMERICA) & (ck % 10 != 0)                       # load.sql:81-82
        c_region = np.where(upd, R_AMERICA, c_region)
        c_nation = np.where(upd, N_UNITED_STATES, c_nation)
        c_digit = np.where(upd, 6, c_digit)
        n_add = 25 * min(100, self.n_c)                                      # load.sql:93-95 (c_custkey <= 100) x 25
        ak = self.n_c + 1 + np.arange(n_add, dtype=np.int64)                 # row_number() + 3000000 at SF100
        self.c_custkey = np.concatenate([ck, ak])
        self.c_region = np.concatenate([c_region, np.full(n_add, R_OCEANIA, dtype=np.int64)])
        self.c_nation = np.concatenate([c_nation, N_OCEANIA0 + (np.arange(n_add, dtype=np.int64) % 25)])
        self.c_digit = np.concatenate([c_digit, np.full(n_add, 1, dtype=np.int64)])
        c_uk15 = (self.c_nation == N_UNITED_KINGDOM) & ((self.c_digit == 1) | (self.c_digit == 5))
        self.c1 = self.c_custkey[self.c_region != R_ASIA]                    # load.sql:98
        self.c2 = self.c_custkey[self.c_region == R_ASIA]                    # :99
        self.c3 = self.c_custkey[c_uk15]                                     # :100
        # ---- supplier ----------------------------------------------------------------------------------
        sk = np.arange(1, self.n_s + 1, dtype=np.int64)
        s_region = _mix(xp, sk, s + 11) % 5
        s_nation = s_region * 5 + _mix(xp, sk, s + 12) % 5
        s_digit = _mix(xp, sk, s + 13) % 10
        upd = (s_region != R_ASIA) & (sk % 10 != 0)                          # load.sql:139-140
        self.s_suppkey = sk
        self.s_region = np.where(upd, R_ASIA, s_region)
        self.s_nation = np.where(upd, N_CHINA, s_nation)
        self.s_digit = np.where(upd, 6, s_digit)
        s_uk15 = (self.s_nation == N_UNITED_KINGDOM) & ((self.s_digit == 1) | (self.s_digit == 5))
        self.s1 = sk[self.s_region == R_ASIA]                                # :142
        self.s2 = sk[self.s_nation == N_UNITED_STATES]                       # :143
        self.s3 = sk[s_uk15]                                                 # :144
        # ---- part --------------------------------------------------------------------------------------
        pk = np.arange(1, self.n_p + 1, dtype=np.int64)
        p_mfgr = 1 + _mix(xp, pk, s + 21) % 5
        p_category = p_mfgr * 10 + 1 + _mix(xp, pk, s + 22) % 5
        p_brand = p_category * 100 + 1 + _mix(xp, pk, s + 23) % 40
        p_size = 1 + _mix(xp, pk, s + 24) % 50
        p_category = np.where((p_category != 12) & (p_size <= 25), 12, p_category)                   # :185
        p_category = np.where((p_category != 14) & (p_category != 12) & (pk % 2 == 0), 14, p_category)  # :186
        p_brand = np.where(pk % 3 == 0, 2239, p_brand)                                               # :187
        self.p_partkey, self.p_mfgr, self.p_category, self.p_brand = pk, p_mfgr, p_category, p_brand
        # ---- date --------------------------------------------------------------------------------------
        self.d_datekey, self.d_year = date_table()
        self.year_views = [self.d_datekey[self.d_year == y].astype(np.int64) for y in range(1992, 1999)]  # :190-196
        # order keys: order o (0-based) has key (o / 8) * 32 + o % 8 + 1 (eight used, twenty-four skipped), 4 lines each
        n_orders = (self.n_lo + 3) // 4
        self.max_orderkey = ((n_orders - 1) // 8) * 32 + (n_orders - 1) % 8 + 1 if n_orders else 0
        f = self.max_orderkey / 600_000_000.0
        self.t400 = int(round(400_000_000 * f))                              # load.sql:110 etc.
        self.bands = [int(round(80_000_000 * f * y)) for y in range(1, 6)]   # :203-231: ends of 1992..1996
        self.t401 = int(round(401_000_000 * f))                              # :237

    def params(self):
        return {"n_lineorder": self.n_lo, "n_customer": len(self.c_custkey), "n_supplier": self.n_s, "n_part": self.n_p,
                "max_orderkey": self.max_orderkey, "orderkey_400M": self.t400, "orderkey_401M": self.t401,
                "year_band_ends": self.bands,
                "view_cardinalities": {"c1": len(self.c1), "c2": len(self.c2), "c3": len(self.c3),
                                       "s1": len(self.s1), "s2": len(self.s2), "s3": len(self.s3)},
                "load_sql_moduli_at_sf100": {"c1": 2942519, "c2": 59981, "c3": 2402, "s1": 183979, "s2": 787, "s3": 191}}

    # ---- lineorder rows [lo, hi): the base row, then load.sql's UPDATEs in order ----------------------------
    def lineorder(self, lo, hi, xp=_NP, cols=("lo_custkey", "lo_suppkey", "lo_partkey", "lo_orderdate"), views=None,
                  row_salt=0):
        """row_salt (a multiple of 4): another lineorder of the same shape -- order keys, skew phases and the rules of
        load.sql as for rows [lo, hi), the per-row random draws those of rows [lo + row_salt, hi + row_salt).  Rank r of
        a weak-scaling run generates its own table with row_salt = r * n_lo: same work, different tuples."""
        s = _wrap(self.seed * 1000003)
        if views is None:
            views = self.device_views(xp)
        i = xp.arange(lo, hi)
        o = i >> 2
        okey = (o >> 3) * 32 + (o & 7) + 1
        ih = i + int(row_salt)
        oh = o + (int(row_salt) >> 2)
        qty = 1 + _mix(xp, ih, s + 31) % 50
        cust = 1 + _mix(xp, oh, s + 32) % self.n_c
        supp = 1 + _mix(xp, ih, s + 33) % self.n_s
        early = okey < self.t400
        late = okey >= self.t400

        def cust_attrs(ck):
            """region / is-UK1-or-UK5 of customer ck after load.sql:81-96 (base rows by formula, added rows OCEANIA)"""
            base = ck <= self.n_c
            r0 = _mix(xp, ck, s + 1) % 5
            upd = (r0 != R_AMERICA) & (ck % 10 != 0)
            region = xp.where(base, xp.where(upd, R_AMERICA, r0), R_OCEANIA)
            nation = r0 * 5 + _mix(xp, ck, s + 2) % 5
            digit = _mix(xp, ck, s + 3) % 10
            uk15 = base & (~upd) & (nation == N_UNITED_KINGDOM) & ((digit == 1) | (digit == 5))
            return region, uk15

        def supp_attrs(sk):
            r0 = _mix(xp, sk, s + 11) % 5
            upd = (r0 != R_ASIA) & (sk % 10 != 0)
            region = xp.where(upd, R_ASIA, r0)
            nation = xp.where(upd, N_CHINA, r0 * 5 + _mix(xp, sk, s + 12) % 5)
            return region, nation

        # customer rules (load.sql:102-132); a rule whose view is empty on this instance changes nothing
        region, _ = cust_attrs(cust)
        if len(self.c1):
            cust = xp.where((region == R_ASIA) & early, xp.take(views["c1"], okey % len(self.c1)), cust)
        region, _ = cust_attrs(cust)
        if len(self.c2):
            cust = xp.where((region == R_AMERICA) & (okey % 3 == 0) & late, xp.take(views["c2"], okey % len(self.c2)), cust)
        region, _ = cust_attrs(cust)
        if len(self.c3):
            cust = xp.where((region == R_AMERICA) & (qty >= 38) & late, xp.take(views["c3"], okey % len(self.c3)), cust)
        # supplier rules (load.sql:146-178)
        sregion, snation = supp_attrs(supp)
        if len(self.s1):
            supp = xp.where((snation == N_UNITED_STATES) & early, xp.take(views["s1"], okey % len(self.s1)), supp)
        sregion, snation = supp_attrs(supp)
        if len(self.s2):
            supp = xp.where((sregion == R_ASIA) & late & (qty <= 6), xp.take(views["s2"], okey % len(self.s2)), supp)
        sregion, snation = supp_attrs(supp)
        if len(self.s3):
            _, c_uk15 = cust_attrs(cust)
            supp = xp.where((sregion == R_ASIA) & (~c_uk15) & (qty >= 43) & early,
                            xp.take(views["s3"], okey % len(self.s3)), supp)
        out = {}
        if "lo_orderkey" in cols:
            out["lo_orderkey"] = xp.cast(okey, np.uint32)
        if "lo_custkey" in cols:
            out["lo_custkey"] = xp.cast(cust, np.uint32)
        if "lo_suppkey" in cols:
            out["lo_suppkey"] = xp.cast(supp, np.uint32)
        if "lo_partkey" in cols:
            out["lo_partkey"] = xp.cast(1 + _mix(xp, ih, s + 34) % self.n_p, np.uint32)
        if "lo_orderdate" in cols:
            # load.sql:198-245: the year is a function of the order-key band, the day of lo_orderkey % 365
            doy = okey % 365
            year_idx = (okey > self.bands[0]) * 1 + (okey > self.bands[1]) * 1 + (okey > self.bands[2]) * 1 + \
                       (okey > self.bands[3]) * 1 + (okey > self.bands[4]) * 1 + (okey > self.t401) * 1
            # the seven views concatenated, 366 slots per year (1992/1996 have 366 days; index 365 is never asked for).
            # 1998 has 364 rows (the table ends on 1998-12-30): for lo_orderkey % 365 = 364 load.sql's scalar
            # subquery finds no row and yields NULL; the slot holds 0 here, which equals no d_datekey either
            out["lo_orderdate"] = xp.cast(xp.take(views["dates"], year_idx * 366 + doy), np.uint32)
        if "lo_quantity" in cols:
            out["lo_quantity"] = xp.cast(qty, np.uint16)
        if "lo_revenue" in cols:
            out["lo_revenue"] = xp.cast(100 + _mix(xp, ih, s + 35) % 9900, np.uint32)
        if "lo_supplycost" in cols:
            out["lo_supplycost"] = xp.cast(50 + _mix(xp, ih, s + 36) % 4950, np.uint32)
        return out

    def device_views(self, xp):
        dates = np.zeros(7 * 366, dtype=np.int64)
        for y, v in enumerate(self.year_views):
            dates[y * 366:y * 366 + len(v)] = v
        return {"c1": xp.asarray(self.c1), "c2": xp.asarray(self.c2), "c3": xp.asarray(self.c3),
                "s1": xp.asarray(self.s1), "s2": xp.asarray(self.s2), "s3": xp.asarray(self.s3),
                "dates": xp.asarray(dates)}

    def lineorder_torch(self, lo, hi, device, cols=("lo_custkey", "lo_suppkey", "lo_partkey", "lo_orderdate"),
                        block=1 << 25, row_salt=0):
        """the same rows as torch tensors on `device`, generated there in blocks (int32 bit patterns of the uint32
        columns)"""
        import torch
        xp = _Torch(device)
        views = self.device_views(xp)
        outs = {c: torch.empty(hi - lo, dtype=torch.int16 if c == "lo_quantity" else torch.int32, device=device)
                for c in cols}
        for b in range(lo, hi, block):
            e = min(hi, b + block)
            part = self.lineorder(b, e, xp, cols, views, row_salt=row_salt)
            for c in cols:
                outs[c][b - lo:e - lo] = part[c]
        return outs

    # ---- the build sides of the SSB-skew queries (filters of benchmark/ssb-skew/queries/*.sql) -------------
    def build_sides(self, query):
        c_keep = {"q4.1": self.c_region == R_AMERICA, "q4.2": self.c_region == R_AMERICA,
                  "q4.3": self.c_region == R_AMERICA, "q3.1": self.c_region == R_ASIA}.get(query)
        s_keep = {"q4.1": self.s_region == R_AMERICA, "q4.2": self.s_region == R_AMERICA,
                  "q4.3": self.s_nation == N_UNITED_STATES, "q3.1": self.s_region == R_ASIA,
                  "q2.1": self.s_region == R_AMERICA}.get(query)
        p_keep = {"q4.1": (self.p_mfgr == 1) | (self.p_mfgr == 2), "q4.2": (self.p_mfgr == 1) | (self.p_mfgr == 2),
                  "q4.3": self.p_category == 14, "q2.1": self.p_category == 12}.get(query)
        d_keep = {"q4.1": np.ones(2556, dtype=bool), "q2.1": np.ones(2556, dtype=bool),
                  "q4.2": (self.d_year == 1997) | (self.d_year == 1998),
                  "q4.3": (self.d_year == 1997) | (self.d_year == 1998),
                  "q3.1": (self.d_year >= 1992) & (self.d_year <= 1997)}[query]
        sides = {}
        if c_keep is not None:
            sides["customer"] = {"keys": self.c_custkey[c_keep].astype(np.uint32),
                                 "payload": {"c_nation": self.c_nation[c_keep].astype(np.uint16)},
                                 "range": (1, int(self.c_custkey[-1])), "probe_col": "lo_custkey"}
        if s_keep is not None:
            sides["supplier"] = {"keys": self.s_suppkey[s_keep].astype(np.uint32),
                                 "payload": {"s_nation": self.s_nation[s_keep].astype(np.uint16)},
                                 "range": (1, self.n_s), "probe_col": "lo_suppkey"}
        if p_keep is not None:
            sides["part"] = {"keys": self.p_partkey[p_keep].astype(np.uint32),
                             "payload": {"p_brand": self.p_brand[p_keep].astype(np.uint16)},
                             "range": (1, self.n_p), "probe_col": "lo_partkey"}
        sides["date"] = {"keys": self.d_datekey[d_keep], "payload": {"d_year": self.d_year[d_keep]},
                         "range": (int(self.d_datekey[0]), int(self.d_datekey[-1])), "probe_col": "lo_orderdate"}
        return sides


# the join order of each query's pipeline = its FROM list without LINEORDER (path 0, what the reference's
# `SET disabled_optimizers TO 'join_order'` pins)
QUERY_JOINS = {"q4.1": ["customer", "supplier", "part", "date"], "q4.2": ["customer", "supplier", "part", "date"],
               "q4.3": ["customer", "supplier", "part", "date"], "q3.1": ["customer", "supplier", "date"],
               "q2.1": ["date", "part", "supplier"]}
PROBE_COLS = ["lo_custkey", "lo_suppkey", "lo_partkey", "lo_orderdate"]


# WHERE clauses of benchmark/ssb-skew/queries/*.sql over the coded dimension attributes (strings are codes here: region
# 1 = AMERICA, 2 = ASIA; nation 9 = UNITED STATES; p_mfgr 1..5; p_category = mfgr * 10 + 1..5)
QUERY_WHERE = {
    "q4.1": {"customer": "c_region = 1", "supplier": "s_region = 1", "part": "(p_mfgr = 1 OR p_mfgr = 2)"},
    "q4.2": {"customer": "c_region = 1", "supplier": "s_region = 1", "part": "(p_mfgr = 1 OR p_mfgr = 2)",
             "date": "(d_year = 1997 OR d_year = 1998)"},
    "q4.3": {"customer": "c_region = 1", "supplier": "s_nation = %d" % N_UNITED_STATES, "part": "p_category = 14",
             "date": "(d_year = 1997 OR d_year = 1998)"},
    "q3.1": {"customer": "c_region = 2", "supplier": "s_region = 2", "date": "d_year >= 1992 AND d_year <= 1997"},
    "q2.1": {"part": "p_category = 12", "supplier": "s_region = 1"},
}
DIM_KEY = {"customer": ("c_custkey", "lo_custkey"), "supplier": ("s_suppkey", "lo_suppkey"),
           "part": ("p_partkey", "lo_partkey"), "date": ("d_datekey", "lo_orderdate")}


def reference_form(inst, query, lineorder_cols):
    """how the same pipeline reads for the reference: full dimension tables with their PRIMARY KEYs (load.sql:21-72)
    and coded attribute columns, the query's filters as WHERE clauses (pushed into the build-side scans), COUNT(*) sink,
    textual join order = QUERY_JOINS (pinned with SET disabled_optimizers TO 'join_order').  node_info: what
    SelSampleEnumeration reads off that plan (base-table cardinality, predicate, unique) -- source first."""
    tables = {"lineorder": dict(lineorder_cols),
              "customer": {"c_custkey": inst.c_custkey.astype(np.uint32), "c_region": inst.c_region.astype(np.uint8),
                           "c_nation": inst.c_nation.astype(np.uint16)},
              "supplier": {"s_suppkey": inst.s_suppkey.astype(np.uint32), "s_region": inst.s_region.astype(np.uint8),
                           "s_nation": inst.s_nation.astype(np.uint16)},
              "part": {"p_partkey": inst.p_partkey.astype(np.uint32), "p_mfgr": inst.p_mfgr.astype(np.uint8),
                       "p_category": inst.p_category.astype(np.uint8)},
              "date": {"d_datekey": inst.d_datekey, "d_year": inst.d_year}}
    names = QUERY_JOINS[query]
    where = QUERY_WHERE[query]
    sql = "SELECT COUNT(*) FROM lineorder"
    for n in names:
        sql += " JOIN %s ON %s = %s" % (n, DIM_KEY[n][1], DIM_KEY[n][0])
    conds = [where[n] for n in names if n in where]
    if conds:
        sql += " WHERE " + " AND ".join(conds)
    n_lo = len(next(iter(lineorder_cols.values())))
    node_info = [(n_lo, False, False)] + [(len(tables[n][DIM_KEY[n][0]]), n in where, True) for n in names]
    return {"tables": {k: v for k, v in tables.items() if k == "lineorder" or k in names},
            "pk": {n: DIM_KEY[n][0] for n in names}, "query": sql,
            "settings": ["SET disabled_optimizers TO 'join_order'"], "node_info": node_info}


def workload(query="q4.1", sf=1.0, seed=1337, n_lo=None, n_c=None, n_s=None, n_p=None, rows=None, host_probe=True):
    """the workload dict of polr_amd.workloads for one SSB-skew query.  rows = (lo, hi): only that contiguous
    partition of lineorder (rank r of N, or the CPU baseline's sample).  host_probe=False: no probe columns are
    generated here -- wl["instance"].lineorder_torch(...) makes them on the device."""
    z = sizes(sf)
    inst = Instance(n_lo if n_lo is not None else z["n_lo"], n_c or z["n_c"], n_s or z["n_s"], n_p or z["n_p"], seed)
    lo, hi = rows if rows is not None else (0, inst.n_lo)
    sides = inst.build_sides(query)
    joins = []
    for name in QUERY_JOINS[query]:
        b = sides[name]
        kmin, kmax = b["range"]
        joins.append({"name": name, "keys": [b["keys"]], "key_names": [{"customer": "c_custkey", "supplier": "s_suppkey",
                                                                          "part": "p_partkey", "date": "d_datekey"}[name]],
                      "payload": b["payload"], "key_src": [(-1, PROBE_COLS.index(b["probe_col"]))],
                      "perfect": (kmin, kmax) if kmax - kmin <= 1_000_000 else None, "key_range": (kmin, kmax)})
    probe = {"name": "lineorder", "rows": (lo, hi)}
    wl = {"name": "ssb_skew_" + query.replace(".", ""), "probe": probe, "joins": joins, "instance": inst,
          "query": query, "params": inst.params()}
    if host_probe:
        probe["cols"] = inst.lineorder(lo, hi, _NP, PROBE_COLS)
        wl["ref"] = reference_form(inst, query, probe["cols"])
    return wl
