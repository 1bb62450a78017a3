import json,sys
d=json.load(open(sys.argv[1])); p=d["val_acc_parity"]
print({k:v for k,v in d.items() if k!="val_acc_parity"}); print(p["settled_device_minus_cpu"], p["ok_settled"], p["lr_replay_same_epochs"])
for r in p["per_seed"]:
  print(r["seed"]); print(" dev acc", [round(x,3) for x in r["device"]["val_acc"]]); print(" cpu acc", [round(x,3) for x in r["cpu"]["val_acc"]]); print(" dev loss", [round(x,3) for x in r["device"]["val_loss"]]); print(" cpu loss", [round(x,3) for x in r["cpu"]["val_loss"]]); print(" train", [round(x,3) for x in r["device"]["train_acc"]], [round(x,3) for x in r["cpu"]["train_acc"]]); print(r["device"]["lr_replay"]["fired_after_epochs"], r["cpu"]["lr_replay"]["fired_after_epochs"], r["device"]["seconds"], r["cpu"]["seconds"])
