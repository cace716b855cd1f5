import sys, torch
a = torch.load(sys.argv[1]); b = torch.load(sys.argv[2])
bad = 0
for i, ((oa, la), (ob, lb)) in enumerate(zip(a, b)):
    if not (torch.equal(oa, ob) and torch.equal(la, lb)):
        bad += 1
        print("differs", i, (oa.float() - ob.float()).abs().max().item(), (la - lb).abs().max().item())
print("compared", len(a), "configs;", bad, "differ")
