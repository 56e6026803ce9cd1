"""tools/search_fuzz.py: the bf16 pre-filter search against the all-f32 search on random shapes around every plan switch (64/65, 768, 1024 queries; 64k-row sample, n/4 >= 64k rows), unit / scaled / duplicated rows, k in {1, 5, 20, 32}: ids and distances must be equal bit for bit."""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd import ops
dev = torch.device("cuda:0")
random.seed(7)
gen = torch.Generator(device=dev).manual_seed(11)
bad = 0
cases = [(262144, 768), (262143, 767), (262208, 769), (300000, 1023), (300000, 1024), (300000, 1025), (70000, 2000), (1000000, 1500),
         (65535, 900), (65600, 64), (65600, 65), (524288, 4096), (262144 * 4 + 7, 777)]
for _ in range(22):
    cases.append((random.choice([random.randint(1, 3000), random.randint(60000, 70000), random.randint(250000, 280000), random.randint(500000, 1200000)]),
                  random.choice([random.randint(1, 70), random.randint(100, 300), random.randint(700, 1100), random.randint(2000, 4100)])))
for n, nq in cases:
    k = random.choice([1, 5, 20, 32])
    mode = random.choice(["unit", "scaled", "dups"])
    db = torch.randn(n, 128, generator=gen, device=dev)
    if mode == "unit":
        db = torch.nn.functional.normalize(db, dim=1)
    elif mode == "scaled":
        db = db * torch.exp(2.0 * torch.randn(n, 1, generator=gen, device=dev))
    else:
        db = torch.nn.functional.normalize(db, dim=1)
        if n > 100:
            db[n // 3: n // 3 + min(3000, n // 4)] = db[n // 3]
    rows = torch.randint(0, n, (nq,), generator=gen, device=dev)
    q = db[rows] + 0.05 * db[rows].norm(dim=1, keepdim=True) / 11.3 * torch.randn(nq, 128, generator=gen, device=dev)
    sq = ops.row_sqnorm(db)
    dbh = ops.rows_to_bf16(db)
    D1, I1 = ops.search_l2(db, sq, q, k, db_bf16=dbh)
    D0, I0 = ops.search_l2(db, sq, q, k)
    ok = torch.equal(I0, I1) and torch.equal(D0, D1)
    if not ok:
        bad += 1
        print("MISMATCH", n, nq, k, mode, int((I0 != I1).sum()), flush=True)
    del db, dbh, sq
print(f"{len(cases)} cases, {bad} mismatches")
